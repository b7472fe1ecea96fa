#!/bin/bash
# ThreadSanitizer pass over the HOST side of libvqvdb_hip.so: the library is rebuilt with -Xarch_host -fsanitize=thread
# (device code unchanged) and tools/tsan_host_driver.cpp hammers the entry points that work without a GPU from 8 threads: creates failing at
# every stage (missing file, bad magic, truncated table, valid pack -> no device), the thread-local error strings, vqhip_multi_create's error
# path.  Build container only (no GPU needed; ~2.5 min).   bash tools/tsan_host_check.sh  ->  "... 0 unexpected results" and no TSAN report
set -e
R=$(cd "$(dirname "$0")/.." && pwd); T=$(mktemp -d /tmp/vqhip_tsan.XXXX)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O1 -g -std=c++17 -ffp-contract=off -fPIC -shared -Xarch_host -fsanitize=thread -Wno-unused-value -Wno-unused-result \
    $R/vqvdb_amd/csrc/vq_runtime.hip -o $T/libvqvdb_hip_tsan.so
/opt/rocm/lib/llvm/bin/clang++ -std=c++17 -O1 -g -fsanitize=thread -I$R/include $R/tools/tsan_host_driver.cpp -o $T/drv -L$T -lvqvdb_hip_tsan \
    -Wl,-rpath,$T -Wl,-rpath,/opt/rocm/lib -L/opt/rocm/lib -lamdhip64 -lpthread
python3 -c "
import sys; sys.path.insert(0, '$R')
from vqvdb_amd import synth, weightpack
open('$T/m.vqw', 'wb').write(weightpack.dumps(synth.make_weights(0)))"
echo "__tsan symbols in the library: $(nm -D $T/libvqvdb_hip_tsan.so | grep -c __tsan)"
TSAN_OPTIONS="halt_on_error=1" $T/drv $T/m.vqw
rm -rf $T
