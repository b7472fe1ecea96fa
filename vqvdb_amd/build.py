"""Builds libvqvdb_hip.so in-tree for gfx950 (hipcc cross-compiles without a GPU).

    python -m vqvdb_amd.build [--report]

-ffp-contract=off: the kernels' arithmetic contract uses explicit fmaf only.
"""
from __future__ import annotations

import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "csrc", "vq_runtime.hip")
# every file of csrc/ is part of the one translation unit (vq_runtime.hip includes the kernel headers and vq_train_full.inc)
DEPS = sorted(os.path.join(HERE, "csrc", f) for f in os.listdir(os.path.join(HERE, "csrc")) if f.endswith((".hip", ".h", ".inc"))) + [
    os.path.join(os.path.dirname(HERE), "include", "vqvdb_hip.h")]
LIB = os.path.join(HERE, "libvqvdb_hip.so")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared",
         "-Wno-unused-value", "-Wno-unused-result"]


def build(force: bool = False, report: bool = False) -> str:
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    stale = force or report or not os.path.exists(LIB) or any(os.path.getmtime(d) > os.path.getmtime(LIB) for d in DEPS)
    if stale:
        cmd = [hipcc, *FLAGS, SRC, "-o", LIB]
        if report:
            cmd.insert(1, "-Rpass-analysis=kernel-resource-usage")
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            sys.stderr.write(r.stderr)
            raise RuntimeError("hipcc failed building libvqvdb_hip.so")
        if report:
            name = None
            for line in r.stderr.splitlines():
                if "Function Name:" in line:
                    name = line.split("Function Name:")[1].split("[")[0].strip()
                for key in ("VGPRs:", "AGPRs:", "ScratchSize", "Occupancy", "VGPRs Spill", "SGPRs Spill", "LDS Size"):
                    if key in line and "remark" in line:
                        print(f"{name:110s} {line.split('remark:')[1].split('[-R')[0].strip()}")
    build_harness(force or stale)
    build_default_embedded_harness(force or stale)
    return LIB


HARNESS = os.path.join(HERE, "host", "leaf_harness")


def build_harness(force: bool = False) -> str:
    """C++ host side: IVQVAECodec adapter + factory + orchestrator-style harness (g++, links the C ABI)."""
    srcs = [os.path.join(HERE, "host", f) for f in ("leaf_harness.cpp", "codec_factory.cpp")]
    deps = srcs + [os.path.join(HERE, "host", "vqvdb_stream.hpp"), os.path.join(HERE, "host", "codec_interface.hpp"),
                   os.path.join(os.path.dirname(HERE), "include", "vqvdb_hip_backend.hpp"),
                   os.path.join(os.path.dirname(HERE), "include", "vqvdb_hip.h")]
    if force or not os.path.exists(HARNESS) or any(os.path.getmtime(d) > os.path.getmtime(HARNESS) for d in deps):
        cmd = ["g++", "-std=c++17", "-O2", "-Wall", "-Wextra", "-o", HARNESS, *srcs, "-L" + HERE, "-lvqvdb_hip", "-Wl,-rpath,$ORIGIN/.."]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            sys.stderr.write(r.stderr)
            raise RuntimeError("g++ failed building leaf_harness")
    return HARNESS


def build_harness_embedded(pack: bytes, exe: str, mode: str = "header", ref_tool: str | None = None) -> str:
    """leaf_harness with a weight pack compiled in, for CodecConfig::source = EmbeddedModel{}.

    mode "header": the adapter includes the generated header (-DVQVDB_HIP_EMBEDDED_PACK_HEADER);
    mode "object": the header is compiled as a C translation unit of its own and linked (-DVQVDB_HIP_EMBEDDED_PACK);
    ref_tool: path of the reference's python/convert_to_header.py — the header is then produced by THAT tool
    (build container only) to show that its output links too."""
    import tempfile
    from . import weightpack
    tmp = tempfile.mkdtemp(prefix="vqhip_embed_")
    try:
        hdr = os.path.join(tmp, "vqhip_pack.h")
        if ref_tool:
            vqw = os.path.join(tmp, "model.vqw")
            with open(vqw, "wb") as f:
                f.write(pack)
            subprocess.run([sys.executable, ref_tool, vqw, hdr, "--name", weightpack.HEADER_SYMBOL], check=True, capture_output=True)
        else:
            with open(hdr, "w") as f:
                f.write(weightpack.to_header(pack, "embedded.vqw"))
        srcs = [os.path.join(HERE, "host", f) for f in ("leaf_harness.cpp", "codec_factory.cpp")]
        cmd = ["g++", "-std=c++17", "-O1", "-Wall", "-Wextra", "-o", exe, *srcs]
        if mode == "header":
            cmd += ["-DVQVDB_HIP_EMBEDDED_PACK_HEADER=\"vqhip_pack.h\"", "-I", tmp]
        else:
            obj = os.path.join(tmp, "vqhip_pack.o")
            subprocess.run(["gcc", "-x", "c", "-c", hdr, "-o", obj], check=True, capture_output=True)
            cmd += ["-DVQVDB_HIP_EMBEDDED_PACK", obj]
        cmd += ["-L" + HERE, "-lvqvdb_hip", "-Wl,-rpath,$ORIGIN/..", "-Wl,-rpath," + HERE]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            sys.stderr.write(r.stderr[-4000:])
            raise RuntimeError("g++ failed building the embedded-pack leaf_harness")
    finally:
        import shutil
        shutil.rmtree(tmp, ignore_errors=True)
    return exe


HARNESS_EMBEDDED = os.path.join(HERE, "host", "leaf_harness_embedded")


def build_default_embedded_harness(force: bool = False) -> str:
    """host/leaf_harness_embedded: the harness with the synthetic seed-0 weight pack compiled in (object mode), so the
    EmbeddedModel route — the one both reference SOPs take — is a built artefact that the CPU and GPU tests run."""
    deps = [os.path.join(HERE, "host", f) for f in ("leaf_harness.cpp", "codec_factory.cpp", "vqvdb_stream.hpp", "codec_interface.hpp")] + [
        os.path.join(os.path.dirname(HERE), "include", "vqvdb_hip_backend.hpp"), os.path.join(os.path.dirname(HERE), "include", "vqvdb_hip.h"),
        os.path.join(HERE, "weightpack.py"), os.path.join(HERE, "synth.py"), LIB]
    if force or not os.path.exists(HARNESS_EMBEDDED) or any(os.path.getmtime(d) > os.path.getmtime(HARNESS_EMBEDDED) for d in deps):
        from . import synth, weightpack
        build_harness_embedded(weightpack.dumps(synth.make_weights(0)), HARNESS_EMBEDDED, "object")
    return HARNESS_EMBEDDED


if __name__ == "__main__":
    build(force=True, report="--report" in sys.argv)
    print(LIB)
