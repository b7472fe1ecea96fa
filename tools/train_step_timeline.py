"""Timeline of ONE full training step from a rocprofv3 kernel trace: per queue (HIP stream) the kernels in start order with their start /
end relative to the step's first kernel, the idle time before each, and the busy / idle totals per queue.
Usage (GPU box):
    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tl -o t -- python tools/train_step_timeline.py --run 2048
    python tools/train_step_timeline.py gpurun_out/tl > gpurun_out/train_step_timeline.txt"""
import csv
import glob
import os
import sys

if len(sys.argv) > 1 and sys.argv[1] == "--run":
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import torch
    from vqvdb_amd import synth, weightpack
    from vqvdb_amd.codec import HipCodec
    from vqvdb_amd.full_training import FullTrainer
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
    codec = HipCodec(weightpack.dumps(synth.make_weights(0)))
    tr = FullTrainer(codec)
    x = torch.rand(n, 512, device="cuda")
    for _ in range(8):
        tr.step(x, want_metrics=False)
    torch.cuda.synchronize()
    codec.close()
    sys.exit(0)

rows = []
for p in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    with open(p) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "0"), r["Kernel_Name"].split("(")[0].replace("void ", "")[:70]))
rows.sort()
# a step starts with pack_leaves_k; take the last complete one
starts = [i for i, r in enumerate(rows) if r[3].startswith("pack_leaves_k")]
a, b = starts[-2], starts[-1]
step = rows[a:b]
t0 = step[0][0]
end = max(r[1] for r in step)
print(f"one step: {len(step)} kernels, {(end - t0) / 1e3:.1f} us from the first start to the last end; next step starts at {(rows[b][0] - t0) / 1e3:.1f} us")
queues = sorted({r[2] for r in step}, key=lambda q: -sum(1 for r in step if r[2] == q))
for q in queues:
    ks = [r for r in step if r[2] == q]
    busy = sum(r[1] - r[0] for r in ks) / 1e3
    print(f"\nqueue {q}: {len(ks)} kernels, busy {busy:.1f} us, first start {(ks[0][0] - t0) / 1e3:.1f} us, last end {(ks[-1][1] - t0) / 1e3:.1f} us")
    prev = None
    for r in ks:
        gap = (r[0] - prev) / 1e3 if prev is not None else 0.0
        print(f"  {(r[0] - t0) / 1e3:8.1f} .. {(r[1] - t0) / 1e3:8.1f}  {(r[1] - r[0]) / 1e3:7.1f} us  idle before {gap:6.1f}  {r[3]}")
        prev = r[1]
