// vqvdb_stream.hpp — OpenVDB-free reader/writer for the `.vqvdb` v3 container the path's
// callers stream through (byte layout: SURVEY.md App. B; reference implementation
// src/Utils/VQVDB_Reader.{hpp,cpp}, which needs openvdb::Coord / Mat4s and is kept as-is in
// the reference tree).  Written from the byte layout, little-endian, packed:
//
//   file   : char magic[5]="VQVDB" | u8 version=3 | u8 numGrids | u32 numEmbeddings | u8 latentDimCount
//   grid   : u32 nameLength | name | f32 transform[16] | u16 latentShape[latentDimCount] | u32 totalBlocks
//   blocks : totalBlocks x { i32 origin[3] | u8 indices[prod(latentShape)] }          (76 B per leaf)
//
// The codebook is NOT in the file (it comes from the backend's model source).
#pragma once
#include <array>
#include <cstdint>
#include <cstring>
#include <fstream>
#include <stdexcept>
#include <string>
#include <vector>

namespace vqvdb {

struct Coord3i {
	int32_t x, y, z;
};
static_assert(sizeof(Coord3i) == 12, "origin is 3 x int32");

struct GridMeta {
	std::string name;
	std::array<float, 16> transform{1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
	std::vector<int64_t> latentShape{4, 4, 4};
	uint32_t numEmbeddings = 256;
	uint64_t totalBlocks = 0;
};

class StreamWriter {
   public:
	explicit StreamWriter(const std::string& path) : f_(path, std::ios::binary | std::ios::trunc) {
		if (!f_) throw std::runtime_error("Cannot open output file: " + path);
		const char zero[12] = {0};
		f_.write(zero, 12);  // header is rewritten by close()
	}
	~StreamWriter() {
		try { close(); } catch (...) {}
	}
	void startGrid(const GridMeta& m) {
		if (numGrids_ == 0) { numEmb_ = m.numEmbeddings; dimCount_ = static_cast<uint8_t>(m.latentShape.size()); }
		else if (m.latentShape.size() != dimCount_) throw std::runtime_error("latent rank differs between grids");
		const uint32_t nameLen = static_cast<uint32_t>(m.name.size());
		put(&nameLen, 4);
		put(m.name.data(), nameLen);
		put(m.transform.data(), 64);
		block_ = 1;
		for (int64_t d : m.latentShape) { const uint16_t v = static_cast<uint16_t>(d); put(&v, 2); block_ *= static_cast<size_t>(d); }
		const uint32_t nb = static_cast<uint32_t>(m.totalBlocks);
		put(&nb, 4);
		++numGrids_;
	}
	// indices: [n][block] uint8 ; origins: [n]
	void writeBatch(const uint8_t* indices, const Coord3i* origins, size_t n) {
		buf_.resize(n * (12 + block_));
		char* p = buf_.data();
		for (size_t i = 0; i < n; ++i) {
			std::memcpy(p, &origins[i], 12);
			std::memcpy(p + 12, indices + i * block_, block_);
			p += 12 + block_;
		}
		put(buf_.data(), buf_.size());
	}
	void endGrid() {}
	void close() {
		if (!f_.is_open()) return;
		f_.seekp(0);
		char h[12];
		std::memcpy(h, "VQVDB", 5);
		h[5] = 3;
		h[6] = static_cast<char>(numGrids_);
		std::memcpy(h + 7, &numEmb_, 4);
		h[11] = static_cast<char>(dimCount_);
		f_.write(h, 12);
		f_.close();
		if (f_.fail()) throw std::runtime_error("Error closing the output file.");
	}

   private:
	void put(const void* p, size_t n) {
		f_.write(static_cast<const char*>(p), static_cast<std::streamsize>(n));
		if (!f_) throw std::runtime_error("Failed to write to .vqvdb file.");
	}
	std::ofstream f_;
	std::vector<char> buf_;
	size_t block_ = 64;
	uint8_t numGrids_ = 0, dimCount_ = 0;
	uint32_t numEmb_ = 0;
};

class StreamReader {
   public:
	explicit StreamReader(const std::string& path) : f_(path, std::ios::binary) {
		if (!f_) throw std::runtime_error("Cannot open input file: " + path);
		char h[12];
		get(h, 12, "Failed to read file header.");
		if (std::memcmp(h, "VQVDB", 5) != 0) throw std::runtime_error("Invalid file magic; not a .vqvdb file.");
		if (static_cast<uint8_t>(h[5]) != 3) throw std::runtime_error("Unsupported .vqvdb version.");
		numGrids_ = static_cast<uint8_t>(h[6]);
		std::memcpy(&numEmb_, h + 7, 4);
		dimCount_ = static_cast<uint8_t>(h[11]);
	}
	bool hasNextGrid() const { return grid_ < numGrids_; }
	GridMeta nextGrid() {
		if (left_ != 0) throw std::runtime_error("previous grid not fully consumed");
		GridMeta m;
		uint32_t nameLen;
		get(&nameLen, 4, "Failed to read grid name length.");
		if (nameLen > (1u << 20)) throw std::runtime_error("Grid name length is implausible (corrupt file).");   // same cap as vqhip_decompress_file
		m.name.resize(nameLen);
		get(m.name.data(), nameLen, "Failed to read grid name.");
		get(m.transform.data(), 64, "Failed to read transform.");
		m.latentShape.clear();
		block_ = 1;
		for (int i = 0; i < dimCount_; ++i) { uint16_t v; get(&v, 2, "Failed to read latent shape."); m.latentShape.push_back(v); block_ *= v; }
		uint32_t nb;
		get(&nb, 4, "File appears truncated, failed to read total block count.");
		m.totalBlocks = nb;
		m.numEmbeddings = numEmb_;
		left_ = nb;
		++grid_;
		return m;
	}
	bool hasNext() const { return left_ > 0; }
	// de-interleaves up to maxBatch chunks; returns the number of leaves read
	size_t nextBatch(size_t maxBatch, std::vector<uint8_t>& indices, std::vector<Coord3i>& origins) {
		const size_t n = static_cast<size_t>(std::min<uint64_t>(left_, maxBatch));
		buf_.resize(n * (12 + block_));
		get(buf_.data(), buf_.size(), "File truncated: incomplete block data.");
		indices.resize(n * block_);
		origins.resize(n);
		const char* p = buf_.data();
		for (size_t i = 0; i < n; ++i) {
			std::memcpy(&origins[i], p, 12);
			std::memcpy(indices.data() + i * block_, p + 12, block_);
			p += 12 + block_;
		}
		left_ -= n;
		return n;
	}
	size_t blockSize() const { return block_; }

   private:
	void get(void* p, size_t n, const char* what) {
		f_.read(static_cast<char*>(p), static_cast<std::streamsize>(n));
		if (static_cast<size_t>(f_.gcount()) != n) throw std::runtime_error(what);
	}
	std::ifstream f_;
	std::vector<char> buf_;
	uint8_t numGrids_ = 0, grid_ = 0, dimCount_ = 0;
	uint32_t numEmb_ = 0;
	size_t block_ = 64;
	uint64_t left_ = 0;
};

}  // namespace vqvdb
