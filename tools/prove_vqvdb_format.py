#!/usr/bin/env python3
"""Framing cross-check of the `.vqvdb` v3 container against the reference's REAL reader / writer (src/Utils/VQVDB_Reader.{hpp,cpp}).

Build container only: needs /root/reference (read-only).  Nothing of it is copied into this repository: the reference's
`VQVDB_Reader.cpp` is compiled where it lies, in a temporary directory that also holds a small stand-in for `<openvdb/Types.h>`
(OpenVDB is not installed here) — `openvdb::Coord` = three int32 and `openvdb::math::Mat4s` = sixteen floats with the members the
reader uses (`identity()`, `asPointer()`, a constructor from `const float*`).  The stand-in has the real types' SIZES and nothing
else, so this proves FRAMING (byte offsets, field widths, record order, header finalisation) and nothing about OpenVDB itself.

  (a) the reference WRITER writes a file (2 grids, ragged batches, non-identity transform) ->
      tests/golden/ref_writer_v3.vqvdb (committed as data with `--write-fixture`; compared byte for byte otherwise);
      vqvdb_amd/vqvdbfile.py parses it and re-serialises it byte-identically;
  (b) the reference READER reads (i) the file written by this repository's C++ `vqvdb::StreamWriter` (`leaf_harness streamtest`
      layout, vqvdb_amd/host/vqvdb_stream.hpp) and (ii) a file in the layout `vqhip_compress_file` / `vqvdbfile.dumps` produce, and
      finds the right names, transforms, latent shapes, block counts, origins and indices.

    python tools/prove_vqvdb_format.py [--write-fixture]      -> exit code 0 and "vqvdb format proof: OK"

Reference: src/Utils/VQVDB_Reader.cpp:58-150 (writer), :168-335 (reader); byte layout SURVEY.md App. B.
"""
import os
import shutil
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_SRC = "/root/reference/src"
FIXTURE = os.path.join(ROOT, "tests", "golden", "ref_writer_v3.vqvdb")

TYPES_H = r'''// stand-in for <openvdb/Types.h>: the SIZES of the two types the .vqvdb reader/writer memcpy, nothing else
#pragma once
#include <cstdint>
#include <cstring>
namespace openvdb {
struct Coord { int32_t v[3]; };
namespace math {
struct Mat4s {
	float m[16];
	Mat4s() { identity(); }
	explicit Mat4s(const float* p) { std::memcpy(m, p, sizeof m); }
	void identity() { for (int i = 0; i < 16; ++i) m[i] = (i % 5 == 0) ? 1.0f : 0.0f; }
	const float* asPointer() const { return m; }
};
}  // namespace math
}  // namespace openvdb
static_assert(sizeof(openvdb::Coord) == 12 && sizeof(openvdb::math::Mat4s) == 64, "real OpenVDB sizes");
'''

# the two grids of the fixture: deterministic contents that any language can regenerate
MAIN = r'''
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>
#include "Utils/VQVDB_Reader.hpp"

static uint8_t idxByte(size_t i) { return static_cast<uint8_t>((i * 2654435761u) >> 24); }
static openvdb::Coord originOf(size_t i) { openvdb::Coord c; c.v[0] = int32_t(i % 37) * 8 - 64; c.v[1] = int32_t((i / 37) % 41) * 8; c.v[2] = -int32_t(i / 1517) * 8; return c; }

static int writeFixture(const char* path) {
	VDBStreamWriter w(path);
	const char* names[2] = {"density", "temperature"};
	const size_t counts[2] = {700, 300}, batches[2] = {128, 97};   // ragged: 700 = 5*128 + 60, 300 = 3*97 + 9
	size_t base = 0;
	for (int g = 0; g < 2; ++g) {
		VQVDBMetadata m;
		m.name = names[g];
		m.numEmbeddings = 256;
		m.latentShape = {4, 4, 4};
		m.totalBlocks = counts[g];
		float t[16];
		for (int i = 0; i < 16; ++i) t[i] = (i % 5 == 0 ? 0.25f * (g + 1) : 0.0f) + (i >= 12 && i < 15 ? 1.5f * (i - 11) * (g + 1) : 0.0f);
		t[15] = 1.0f;
		m.transform = openvdb::math::Mat4s(t);
		w.startGrid(m);
		for (size_t s = 0; s < counts[g]; s += batches[g]) {
			const size_t n = std::min(batches[g], counts[g] - s);
			Tensor enc;
			enc.shape = {int64_t(n), 4, 4, 4};
			enc.dtype = DataType::UINT8;
			enc.buffer.resize(n * 64);
			std::vector<openvdb::Coord> org(n);
			for (size_t i = 0; i < n; ++i) {
				org[i] = originOf(base + s + i);
				for (size_t k = 0; k < 64; ++k) enc.buffer[i * 64 + k] = std::byte(idxByte((base + s + i) * 64 + k));
			}
			w.writeBatch(enc, org);
		}
		w.endGrid();
		base += counts[g];
	}
	w.close();
	return 0;
}

// the reference reader over a file of the same logical content, whoever wrote it
static int readCheck(const char* path, size_t maxBatch) {
	VDBStreamReader r(path);
	const char* names[2] = {"density", "temperature"};
	const size_t counts[2] = {700, 300};
	size_t base = 0;
	int g = 0;
	while (r.hasNextGrid()) {
		const VQVDBMetadata m = r.nextGridMetadata();
		if (g > 1 || m.name != names[g] || m.totalBlocks != counts[g] || m.numEmbeddings != 256) return 10 + g;
		if (m.latentShape != std::vector<int64_t>{4, 4, 4}) return 20 + g;
		if (m.transform.asPointer()[0] != 0.25f * (g + 1) || m.transform.asPointer()[13] != 3.0f * (g + 1) || m.transform.asPointer()[15] != 1.0f) return 30 + g;
		size_t seen = 0;
		while (r.hasNext()) {
			EncodedBatch b = r.nextBatch(maxBatch);
			const size_t n = b.origins.size();
			if (n == 0) break;
			if (b.data.dtype != DataType::UINT8 || b.data.shape.size() != 4 || size_t(b.data.shape[0]) != n) return 40 + g;
			for (size_t i = 0; i < n; ++i) {
				const openvdb::Coord o = originOf(base + seen + i);
				if (std::memcmp(&o, &b.origins[i], 12) != 0) return 50 + g;
				for (size_t k = 0; k < 64; ++k)
					if (uint8_t(b.data.buffer[i * 64 + k]) != idxByte((base + seen + i) * 64 + k)) return 60 + g;
			}
			seen += n;
		}
		if (seen != counts[g]) return 70 + g;
		base += counts[g];
		++g;
	}
	std::printf("reference reader: %s -> %d grids, %zu leaves, all fields as written\n", path, g, base);
	return g == 2 ? 0 : 80;
}

int main(int argc, char** argv) {
	try {
		if (argc == 3 && std::string(argv[1]) == "write") return writeFixture(argv[2]);
		if (argc == 4 && std::string(argv[1]) == "read") return readCheck(argv[2], std::stoul(argv[3]));
	} catch (const std::exception& e) {
		std::fprintf(stderr, "exception: %s\n", e.what());
		return 2;
	}
	return 3;
}
'''

# the same logical content through THIS repository's C++ StreamWriter (vqvdb_amd/host/vqvdb_stream.hpp)
OURS = r'''
#include <algorithm>
#include <cstdio>
#include <vector>
#include "vqvdb_stream.hpp"
static uint8_t idxByte(size_t i) { return static_cast<uint8_t>((i * 2654435761u) >> 24); }
int main(int, char** argv) {
	vqvdb::StreamWriter w(argv[1]);
	const char* names[2] = {"density", "temperature"};
	const size_t counts[2] = {700, 300}, batches[2] = {211, 64};
	size_t base = 0;
	for (int g = 0; g < 2; ++g) {
		vqvdb::GridMeta m;
		m.name = names[g];
		m.totalBlocks = counts[g];
		for (int i = 0; i < 16; ++i) m.transform[i] = (i % 5 == 0 ? 0.25f * (g + 1) : 0.0f) + (i >= 12 && i < 15 ? 1.5f * (i - 11) * (g + 1) : 0.0f);
		m.transform[15] = 1.0f;
		w.startGrid(m);
		for (size_t s = 0; s < counts[g]; s += batches[g]) {
			const size_t n = std::min(batches[g], counts[g] - s);
			std::vector<uint8_t> idx(n * 64);
			std::vector<vqvdb::Coord3i> org(n);
			for (size_t i = 0; i < n; ++i) {
				const size_t j = base + s + i;
				org[i] = vqvdb::Coord3i{int32_t(j % 37) * 8 - 64, int32_t((j / 37) % 41) * 8, -int32_t(j / 1517) * 8};
				for (size_t k = 0; k < 64; ++k) idx[i * 64 + k] = idxByte(j * 64 + k);
			}
			w.writeBatch(idx.data(), org.data(), n);
		}
		w.endGrid();
		base += counts[g];
	}
	w.close();
	return 0;
}
'''


def fixture_content():
    """The fixture's logical content, regenerated in numpy (tests/test_host_logic.py uses the same function)."""
    j = np.arange(1000, dtype=np.int64)
    org = np.stack([(j % 37) * 8 - 64, ((j // 37) % 41) * 8, -(j // 1517) * 8], axis=1).astype(np.int32)
    i = np.arange(1000 * 64, dtype=np.uint64)
    idx = (((i * np.uint64(2654435761)) & np.uint64(0xFFFFFFFF)) >> np.uint64(24)).astype(np.uint8).reshape(1000, 64)
    tr = []
    for g in range(2):
        t = np.zeros(16, dtype=np.float32)
        t[[0, 5, 10]] = 0.25 * (g + 1)
        t[12:15] = 1.5 * np.arange(1, 4) * (g + 1)
        t[15] = 1.0
        tr.append(t)
    return [("density", org[:700], idx[:700], tr[0]), ("temperature", org[700:], idx[700:], tr[1])]


def main() -> int:
    if not os.path.isdir(REF_SRC):
        print("vqvdb format proof: SKIPPED (no /root/reference here — this script runs in the build container only)")
        return 0
    sys.path.insert(0, ROOT)
    from vqvdb_amd import vqvdbfile
    tmp = tempfile.mkdtemp(prefix="vqvdb_format_")
    try:
        os.makedirs(os.path.join(tmp, "stand_in", "openvdb"))
        open(os.path.join(tmp, "stand_in", "openvdb", "Types.h"), "w").write(TYPES_H)
        open(os.path.join(tmp, "main.cpp"), "w").write(MAIN)
        open(os.path.join(tmp, "ours.cpp"), "w").write(OURS)
        exe, ours = os.path.join(tmp, "refio"), os.path.join(tmp, "ours")
        r = subprocess.run(["g++", "-std=c++17", "-O1", "-I", os.path.join(tmp, "stand_in"), "-I", REF_SRC, os.path.join(tmp, "main.cpp"),
                            os.path.join(REF_SRC, "Utils", "VQVDB_Reader.cpp"), "-o", exe], capture_output=True, text=True)
        if r.returncode != 0:
            print(r.stderr)
            print("vqvdb format proof: FAILED (the reference reader/writer did not compile against the stand-in)")
            return 1
        r = subprocess.run(["g++", "-std=c++17", "-O1", "-I", os.path.join(ROOT, "vqvdb_amd", "host"), os.path.join(tmp, "ours.cpp"), "-o", ours],
                           capture_output=True, text=True)
        if r.returncode != 0:
            print(r.stderr)
            print("vqvdb format proof: FAILED (vqvdb_stream.hpp writer did not compile)")
            return 1
        # (a) reference writer -> fixture
        ref_file = os.path.join(tmp, "ref.vqvdb")
        if subprocess.run([exe, "write", ref_file]).returncode != 0:
            print("vqvdb format proof: FAILED (reference writer)")
            return 1
        data = open(ref_file, "rb").read()
        if "--write-fixture" in sys.argv:
            open(FIXTURE, "wb").write(data)
            print(f"wrote {FIXTURE} ({len(data)} bytes)")
        elif not os.path.exists(FIXTURE) or open(FIXTURE, "rb").read() != data:
            print("vqvdb format proof: FAILED (tests/golden/ref_writer_v3.vqvdb differs from what the reference writer writes; --write-fixture regenerates)")
            return 1
        grids = vqvdbfile.loads(data)
        want = fixture_content()
        ok = len(grids) == 2 and vqvdbfile.dumps(grids) == data
        for g, (name, org, idx, tr) in zip(grids, want):
            ok = ok and g.name == name and np.array_equal(g.origins, org) and np.array_equal(g.indices, idx) and np.array_equal(g.transform, tr) \
                and tuple(g.latent_shape) == (4, 4, 4)
        print(f"(a) reference writer -> {len(data)} bytes; vqvdbfile parses every field and re-serialises byte-identically: {ok}")
        if not ok:
            print("vqvdb format proof: FAILED (a)")
            return 1
        # (b) reference reader <- our writers
        ours_file, np_file = os.path.join(tmp, "ours.vqvdb"), os.path.join(tmp, "numpy.vqvdb")
        if subprocess.run([ours, ours_file]).returncode != 0:
            print("vqvdb format proof: FAILED (vqvdb::StreamWriter)")
            return 1
        same = open(ours_file, "rb").read() == data
        print(f"(b) vqvdb::StreamWriter (batches of 211 / 64) writes the reference writer's bytes: {same}")
        vqvdbfile.save(np_file, [vqvdbfile.Grid(n, o, i, t) for n, o, i, t in want])          # = the layout vqhip_compress_file leaves
        same_np = open(np_file, "rb").read() == data
        print(f"(b) vqvdbfile.dumps writes the reference writer's bytes: {same_np}")
        for path in (ours_file, np_file, ref_file):
            for mb in (97, 1000, 4096):
                r = subprocess.run([exe, "read", path, str(mb)], capture_output=True, text=True)
                if r.returncode != 0:
                    print(r.stdout, r.stderr)
                    print(f"vqvdb format proof: FAILED (reference reader on {os.path.basename(path)}, batch {mb}: code {r.returncode})")
                    return 1
            print("    " + r.stdout.strip().replace(tmp + "/", ""))
        ok = same and same_np
        print("vqvdb format proof: " + ("OK" if ok else "FAILED"))
        return 0 if ok else 1
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


if __name__ == "__main__":
    sys.exit(main())
