#!/usr/bin/env python3
"""Compile proof of INTEGRATION.md §2 against the reference's REAL factory (src/core/IVQVAECodec.{hpp,cpp}).

Build container only: needs /root/reference (read-only; nothing of it is copied into this repository — the two files are copied to
a temporary directory, patched there exactly as INTEGRATION.md §2 says, compiled with -DENABLE_HIP_BACKEND together with the
adapter include/vqvdb_hip_backend.hpp, linked against vqvdb_amd/libvqvdb_hip.so, and the temporary directory is deleted).

Checks:
  1. the patched reference factory + HipBackend compile and link (g++ -std=c++17 -Wall -Wextra, no warnings from our header);
  2. IVQVAECodec::create(cfg{CUDA, "/nonexistent/model.vqw"}, BackendType::HIP) returns nullptr and prints the reference's
     "Failed to create VQ-VAE backend: ..." line on stderr (IVQVAECodec.cpp:106-109) — no GPU needed: the pack is opened first;
  3. BackendType::LibTorch / ::ONNX keep their numeric values (0, 1) and HIP is 2;
  4. create(cfg, BackendType::ONNX) in this build (ONNX disabled) still takes the reference's default branch -> nullptr.

    python tools/prove_integration.py        -> exit code 0 and "integration proof: OK"
"""
import os
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference/src/core"

MAIN = r'''
#include <cstdio>
#include "core/IVQVAECodec.hpp"
static_assert(static_cast<int>(BackendType::LibTorch) == 0 && static_cast<int>(BackendType::ONNX) == 1 && static_cast<int>(BackendType::HIP) == 2,
              "appending HIP must not renumber the existing backends");
int main() {
	CodecConfig cfg;
	cfg.device = CodecConfig::Device::CUDA;
	cfg.source = std::filesystem::path("/nonexistent/model.vqw");
	std::unique_ptr<IVQVAECodec> a = IVQVAECodec::create(cfg, BackendType::HIP);
	std::unique_ptr<IVQVAECodec> b = IVQVAECodec::create(cfg, BackendType::ONNX);
	cfg.device = CodecConfig::Device::CPU;
	std::unique_ptr<IVQVAECodec> c = IVQVAECodec::create(cfg, BackendType::HIP);
	std::printf("hip_missing_pack=%s onnx_disabled=%s hip_cpu_device=%s\n", a ? "object" : "nullptr", b ? "object" : "nullptr", c ? "object" : "nullptr");
	return (a || b || c) ? 1 : 0;
}
'''


def patch_sources(src_dir: str) -> None:
    hpp = os.path.join(src_dir, "core", "IVQVAECodec.hpp")
    cpp = os.path.join(src_dir, "core", "IVQVAECodec.cpp")
    h = open(hpp).read()
    old = "enum class BackendType { LibTorch, ONNX };"
    assert h.count(old) == 1, "INTEGRATION.md §2 anchor (BackendType enum) not found in the reference header"
    open(hpp, "w").write(h.replace(old, "enum class BackendType { LibTorch, ONNX, HIP };"))
    c = open(cpp).read()
    inc_anchor = "#ifdef ENABLE_TORCH_BACKEND\n#include \"backends/torch/TorchBackend.hpp\"\n#endif\n"
    assert c.count(inc_anchor) == 1, "INTEGRATION.md §2 anchor (backend includes) not found in the reference factory"
    c = c.replace(inc_anchor, inc_anchor + "\n#ifdef ENABLE_HIP_BACKEND\n#include \"backends/hip/HipBackend.hpp\"\n#endif\n")
    case_anchor = "\t\t\tdefault:\n\t\t\t\tthrow std::runtime_error(\n\t\t\t\t    \"Requested backend type is not available"
    assert c.count(case_anchor) == 1, "INTEGRATION.md §2 anchor (factory switch default) not found in the reference factory"
    c = c.replace(case_anchor, "#ifdef ENABLE_HIP_BACKEND\n\t\t\tcase BackendType::HIP:\n\t\t\t\treturn std::unique_ptr<IVQVAECodec>(new HipBackend(config));\n#endif\n" + case_anchor)
    open(cpp, "w").write(c)


def main() -> int:
    if not os.path.isdir(REF):
        print("integration proof: SKIPPED (no /root/reference here — this script runs in the build container only)")
        return 0
    lib = os.path.join(ROOT, "vqvdb_amd", "libvqvdb_hip.so")
    if not os.path.exists(lib):
        from vqvdb_amd.build import build
        build()
    tmp = tempfile.mkdtemp(prefix="vqhip_integration_")
    try:
        src = os.path.join(tmp, "src")
        os.makedirs(os.path.join(src, "core"))
        os.makedirs(os.path.join(src, "backends", "hip"))
        for f in ("IVQVAECodec.hpp", "IVQVAECodec.cpp"):
            shutil.copy(os.path.join(REF, f), os.path.join(src, "core", f))
        shutil.copy(os.path.join(ROOT, "include", "vqvdb_hip.h"), os.path.join(src, "backends", "hip", "vqvdb_hip.h"))
        shutil.copy(os.path.join(ROOT, "include", "vqvdb_hip_backend.hpp"), os.path.join(src, "backends", "hip", "HipBackend.hpp"))
        patch_sources(src)
        open(os.path.join(tmp, "main.cpp"), "w").write(MAIN)
        exe = os.path.join(tmp, "proof")
        cmd = ["g++", "-std=c++17", "-O1", "-Wall", "-Wextra", "-DENABLE_HIP_BACKEND", "-I", src, "-I", os.path.join(src, "core"),
               os.path.join(src, "core", "IVQVAECodec.cpp"), os.path.join(tmp, "main.cpp"), "-o", exe,
               "-L" + os.path.dirname(lib), "-lvqvdb_hip", "-Wl,-rpath," + os.path.dirname(lib)]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            print(r.stderr)
            print("integration proof: FAILED (compile/link)")
            return 1
        ours = [ln for ln in r.stderr.splitlines() if "HipBackend.hpp" in ln or "vqvdb_hip.h" in ln]
        if ours:
            print("\n".join(ours))
            print("integration proof: FAILED (warnings from the adapter headers)")
            return 1
        env = dict(os.environ, LD_LIBRARY_PATH="/opt/rocm/lib:" + os.environ.get("LD_LIBRARY_PATH", ""))
        r = subprocess.run([exe], capture_output=True, text=True, env=env)
        print(r.stdout.strip())
        print("stderr of the patched reference factory:\n  " + "\n  ".join(r.stderr.strip().splitlines()))
        ok = (r.returncode == 0 and "hip_missing_pack=nullptr onnx_disabled=nullptr hip_cpu_device=nullptr" in r.stdout
              and r.stderr.count("Failed to create VQ-VAE backend:") == 3 and "Model file not found at path: /nonexistent/model.vqw" in r.stderr)
        print("integration proof: " + ("OK" if ok else "FAILED"))
        return 0 if ok else 1
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


if __name__ == "__main__":
    sys.path.insert(0, ROOT)
    sys.exit(main())
