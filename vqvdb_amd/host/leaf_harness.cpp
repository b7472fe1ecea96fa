// leaf_harness.cpp — drives the HIP backend exactly the way the reference orchestrator does,
// without OpenVDB/TBB: IVQVAECodec::create -> per batch { pack leaves into a fresh contiguous
// buffer -> TensorView -> encode -> writeBatch }  (VQVAECodec::compress, VQVAECodec.cpp:78-134) and
// { nextBatch -> TensorView -> decode -> copy each 2 KiB leaf out } (::decompress, :137-208).
// Leaves come from / go to raw float32 files so Python tests can compare with the C-ABI path.
//
//   leaf_harness compress   <pack> <leaves.f32> <out.vqvdb> <batch>
//   leaf_harness decompress <pack> <in.vqvdb>   <out.f32>   <batch>
//   leaf_harness compress_stream   <pack> <leaves.f32> <out.vqvdb> <batch>   (vqhip_compress_file: overlapped pipeline)
//   leaf_harness decompress_stream <pack> <in.vqvdb>   <out.f32>   <batch>   (vqhip_decompress_file)
//   leaf_harness loopbench  <pack> <n_leaves> <tmp.vqvdb> <batch>[,<batch>...] [threads]   (both loops, timed per phase, synthetic leaves;
//                           threads: pack loop and leaf copies on that many threads, 0 = half the cores, like the reference's tbb::parallel_for)
//   leaf_harness loopbench_ptrs <pack> <n_leaves> <tmp.vqvdb> <batch>[,<batch>...]   (the loops on vqhip_encode_leaves / vqhip_decode_leaves over
//                           scattered 2 KiB heap blocks: INTEGRATION.md §6)
//   leaf_harness errors     <pack>
//   leaf_harness threads    <pack> <iterations>  (two HipBackend objects driven from two caller threads + create/destroy churn)
//   <pack> = @embedded selects CodecConfig::source = EmbeddedModel{} (builds with -DVQVDB_HIP_EMBEDDED_PACK[_HEADER])
//   leaf_harness streamtest <tmp.vqvdb>          (no GPU needed)
//   leaf_harness readcheck  <ref_writer_v3.vqvdb> <batch>   (no GPU needed: StreamReader over the file the reference's writer wrote)
//   leaf_harness makefile   <out.vqvdb> <n_leaves> (synthetic indices; config-3 input)
#define VQVDB_HIP_STANDALONE
#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>
#include <cstdio>
#include <cstring>
#include <iostream>
#include <unordered_map>

#include "../../include/vqvdb_hip_backend.hpp"
#include "vqvdb_stream.hpp"

namespace {
constexpr size_t LEAF_VOXELS = 512;

std::vector<float> readFloats(const std::string& path) {
	std::ifstream f(path, std::ios::binary | std::ios::ate);
	if (!f) throw std::runtime_error("cannot open " + path);
	const size_t bytes = static_cast<size_t>(f.tellg());
	std::vector<float> v(bytes / sizeof(float));
	f.seekg(0);
	f.read(reinterpret_cast<char*>(v.data()), static_cast<std::streamsize>(v.size() * sizeof(float)));
	return v;
}

vqvdb::Coord3i originOf(size_t i) {  // synthetic leaf origins on the 8-voxel lattice
	return {static_cast<int32_t>(8 * (i % 1024)), static_cast<int32_t>(8 * ((i / 1024) % 1024)), static_cast<int32_t>(8 * (i / 1048576))};
}

std::unique_ptr<IVQVAECodec> makeBackend(const std::string& pack) {
	CodecConfig cfg;
	cfg.device = CodecConfig::Device::CUDA;
	if (pack == "@embedded")
		cfg.source = EmbeddedModel{};  // what both SOPs pass (SOP_VQVDB_Encoder.cpp:63-67); needs a build with an embedded pack
	else
		cfg.source = std::filesystem::path(pack);
	auto be = IVQVAECodec::create(cfg, BackendType::HIP);
	if (!be) throw std::runtime_error("VQVAECodec: Backend cannot be null.");  // VQVAECodec.cpp:71-75
	return be;
}

int compress(const std::string& pack, const std::string& in, const std::string& out, size_t batch) {
	auto backend = makeBackend(pack);
	const std::vector<float> all = readFloats(in);
	const size_t total = all.size() / LEAF_VOXELS;
	const auto t0 = std::chrono::high_resolution_clock::now();
	vqvdb::StreamWriter writer(out);
	vqvdb::GridMeta meta;
	meta.name = "density";
	meta.latentShape = backend->getLatentShape();
	meta.totalBlocks = total;
	writer.startGrid(meta);
	for (size_t start = 0; start < total; start += batch) {
		const size_t B = std::min(batch, total - start);
		std::vector<float> hostData(B * LEAF_VOXELS);  // a fresh buffer per batch, like nextBatch()
		std::vector<vqvdb::Coord3i> origins(B);
		for (size_t i = 0; i < B; ++i) {
			origins[i] = originOf(start + i);
			std::memcpy(hostData.data() + i * LEAF_VOXELS, all.data() + (start + i) * LEAF_VOXELS, LEAF_VOXELS * sizeof(float));
		}
		TensorView view;
		view.data = hostData.data();
		view.shape = {static_cast<int64_t>(B), 1, 8, 8, 8};
		view.dtype = DataType::FLOAT32;
		const Tensor encoded = backend->encode(view);
		if (encoded.dtype != DataType::UINT8 || encoded.shape.size() != 4 || encoded.shape[0] != static_cast<int64_t>(B))
			throw std::runtime_error("unexpected encode result shape");
		writer.writeBatch(encoded.getData<uint8_t>(), origins.data(), B);
	}
	writer.endGrid();
	writer.close();
	const auto ms = std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::high_resolution_clock::now() - t0).count();
	std::printf("Grid Compression Complete in %lld ms (%zu leaves, batch %zu).\n", static_cast<long long>(ms), total, batch);
	return 0;
}

int decompress(const std::string& pack, const std::string& in, const std::string& out, size_t batch) {
	auto backend = makeBackend(pack);
	const auto t0 = std::chrono::high_resolution_clock::now();
	vqvdb::StreamReader reader(in);
	std::ofstream of(out, std::ios::binary | std::ios::trunc);
	size_t leafNo = 0;
	while (reader.hasNextGrid()) {
		const vqvdb::GridMeta meta = reader.nextGrid();
		std::vector<uint8_t> idx;
		std::vector<vqvdb::Coord3i> origins;
		while (reader.hasNext()) {
			const size_t B = reader.nextBatch(batch, idx, origins);
			if (B == 0) break;
			TensorView view;
			view.data = idx.data();
			view.shape = {static_cast<int64_t>(B)};
			view.shape.insert(view.shape.end(), meta.latentShape.begin(), meta.latentShape.end());  // verbatim from the file
			view.dtype = DataType::UINT8;
			const Tensor decoded = backend->decode(view);
			if (decoded.shape.size() != 5) throw std::runtime_error("decode result is not 5-D");
			const float* src = decoded.getData<float>();
			for (size_t i = 0; i < B; ++i) {  // stand-in for touchLeaf + memcpy + setValuesOn
				const vqvdb::Coord3i o = originOf(leafNo + i);
				if (std::memcmp(&o, &origins[i], 12) != 0) throw std::runtime_error("origin mismatch in stream");
				of.write(reinterpret_cast<const char*>(src + i * LEAF_VOXELS), LEAF_VOXELS * sizeof(float));
			}
			leafNo += B;
		}
	}
	const auto ms = std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::high_resolution_clock::now() - t0).count();
	std::printf("Multi-Grid Decompression Complete in %lld ms (%zu leaves, batch %zu).\n", static_cast<long long>(ms), leafNo, batch);
	return 0;
}

// ---- whole-file entry points of the C ABI (read || decode || leaf insert overlapped inside the library) ----
struct CodecHandle {
	vqhip_codec* h = nullptr;
	explicit CodecHandle(const std::string& pack) {
		if (vqhip_create(pack.c_str(), nullptr, 0, 0, &h) != VQHIP_OK) throw std::runtime_error(vqhip_last_error(nullptr));
	}
	void reserve(size_t batch) {  // one-time allocations out of the timed call, like model loading in create()
		if (vqhip_reserve(h, static_cast<int64_t>(batch)) != VQHIP_OK) throw std::runtime_error(vqhip_last_error(h));
	}
	~CodecHandle() { vqhip_destroy(h); }
};

void printStats(const char* what, const vqhip_stream_stats& st, size_t batch) {
	std::printf("%s: %lld leaves in %d grid(s), %.1f ms wall = %.3f M leaves/s (batch %zu; file+framing %.1f ms, leaf alloc %.1f ms, "
	            "gather/scatter %.1f ms, pipeline waited for reader %.1f ms)\n",
	            what, static_cast<long long>(st.leaves), st.grids, st.wall_s * 1e3, st.leaves / st.wall_s / 1e6, batch, st.read_s * 1e3,
	            st.alloc_s * 1e3, st.copy_s * 1e3, st.io_wait_s * 1e3);
}

int compressStream(const std::string& pack, const std::string& in, const std::string& out, size_t batch) {
	CodecHandle codec(pack);
	codec.reserve(batch);
	const std::vector<float> all = readFloats(in);
	const size_t total = all.size() / LEAF_VOXELS;
	std::vector<const float*> ptrs(total);
	std::vector<int32_t> origins(total * 3);
	for (size_t i = 0; i < total; ++i) {
		ptrs[i] = all.data() + i * LEAF_VOXELS;  // stand-in for leaf.buffer().data()
		const vqvdb::Coord3i o = originOf(i);
		origins[3 * i] = o.x, origins[3 * i + 1] = o.y, origins[3 * i + 2] = o.z;
	}
	vqhip_grid_source g{};
	g.name = "density";
	g.transform = nullptr;
	g.leaf_ptrs = ptrs.data();
	g.origins = origins.data();
	g.n_leaves = static_cast<int64_t>(total);
	vqhip_stream_stats st;
	if (vqhip_compress_file(codec.h, out.c_str(), &g, 1, static_cast<int64_t>(batch), &st) != VQHIP_OK) throw std::runtime_error(vqhip_last_error(codec.h));
	printStats("compress_stream", st, batch);
	return 0;
}

// Leaf store standing in for an OpenVDB tree: touchLeaf = hash-map insert keyed by origin + a 2 KiB leaf buffer.
struct LeafStore {
	struct Key {
		int32_t x, y, z;
		bool operator==(const Key& o) const { return x == o.x && y == o.y && z == o.z; }
	};
	struct Hash {
		size_t operator()(const Key& k) const { return (static_cast<size_t>(static_cast<uint32_t>(k.x)) * 73856093u) ^ (static_cast<size_t>(static_cast<uint32_t>(k.y)) * 19349663u) ^ (static_cast<size_t>(static_cast<uint32_t>(k.z)) * 83492791u); }
	};
	std::unordered_map<Key, float*, Hash> leaves;
	std::vector<float*> order;  // file order, for the output file
	std::vector<std::unique_ptr<float[]>> slabs;
	static int beginGrid(void* user, const vqhip_grid_info* g) {  // openvdb::FloatGrid::create + setName/setTransform in the real caller
		LeafStore& self = *static_cast<LeafStore*>(user);
		self.leaves.reserve(self.leaves.size() + g->total_blocks);
		self.order.reserve(self.order.size() + g->total_blocks);
		return 0;
	}
	static int alloc(void* user, int, const int32_t* origins, int64_t n, float** out) {
		LeafStore& self = *static_cast<LeafStore*>(user);
		self.slabs.emplace_back(new float[static_cast<size_t>(n) * LEAF_VOXELS]);
		float* base = self.slabs.back().get();
		for (int64_t i = 0; i < n; ++i) {
			float* leaf = base + i * LEAF_VOXELS;
			self.leaves[Key{origins[3 * i], origins[3 * i + 1], origins[3 * i + 2]}] = leaf;
			self.order.push_back(leaf);
			out[i] = leaf;
		}
		return 0;
	}
};

int decompressStream(const std::string& pack, const std::string& in, const std::string& out, size_t batch) {
	CodecHandle codec(pack);
	codec.reserve(batch);
	LeafStore store;
	vqhip_stream_stats st;
	if (vqhip_decompress_file(codec.h, in.c_str(), static_cast<int64_t>(batch), &LeafStore::beginGrid, &LeafStore::alloc, &store, &st) != VQHIP_OK)
		throw std::runtime_error(vqhip_last_error(codec.h));
	printStats("decompress_stream", st, batch);
	if (store.leaves.size() != store.order.size()) throw std::runtime_error("duplicate origins in stream");
	if (out != "/dev/null") {
		std::ofstream of(out, std::ios::binary | std::ios::trunc);
		for (const float* leaf : store.order) of.write(reinterpret_cast<const char*>(leaf), LEAF_VOXELS * sizeof(float));
	}
	return 0;
}

// synthetic single-grid .vqvdb with pseudo-random indices (decode cost is data-independent): BASELINE config 3 input
int makefile(const std::string& path, size_t total) {
	vqvdb::StreamWriter w(path);
	vqvdb::GridMeta m;
	m.name = "density";
	m.totalBlocks = total;
	w.startGrid(m);
	const size_t B = 65536;
	std::vector<uint8_t> idx(B * 64);
	std::vector<vqvdb::Coord3i> org(B);
	uint32_t x = 2463534242u;
	for (size_t s = 0; s < total; s += B) {
		const size_t n = std::min(B, total - s);
		for (size_t i = 0; i < n * 64; ++i) { x ^= x << 13; x ^= x >> 17; x ^= x << 5; idx[i] = static_cast<uint8_t>(x >> 11); }
		for (size_t i = 0; i < n; ++i) org[i] = originOf(s + i);
		w.writeBatch(idx.data(), org.data(), n);
	}
	w.endGrid();
	w.close();
	std::printf("makefile: %zu leaves -> %s\n", total, path.c_str());
	return 0;
}

// .vqvdb framing round trip without any backend (CPU-only test hook): two grids, ragged batches
int streamtest(const std::string& path) {
	std::vector<uint8_t> idx(1000 * 64);
	for (size_t i = 0; i < idx.size(); ++i) idx[i] = static_cast<uint8_t>((i * 2654435761u) >> 24);
	std::vector<vqvdb::Coord3i> org(1000);
	for (size_t i = 0; i < org.size(); ++i) org[i] = originOf(i);
	{
		vqvdb::StreamWriter w(path);
		for (int g = 0; g < 2; ++g) {
			vqvdb::GridMeta m;
			m.name = g ? "temperature" : "density";
			m.totalBlocks = g ? 300 : 700;
			m.transform[0] = 0.5f + g;
			w.startGrid(m);
			const size_t base = g ? 700 : 0, n = m.totalBlocks;
			for (size_t s = 0; s < n; s += 128) w.writeBatch(idx.data() + (base + s) * 64, org.data() + base + s, std::min<size_t>(128, n - s));
			w.endGrid();
		}
	}
	vqvdb::StreamReader r(path);
	size_t seen = 0;
	int grids = 0;
	while (r.hasNextGrid()) {
		const vqvdb::GridMeta m = r.nextGrid();
		if (m.latentShape != std::vector<int64_t>{4, 4, 4} || m.numEmbeddings != 256 || m.transform[0] != 0.5f + grids) return 1;
		if (m.name != (grids ? "temperature" : "density")) return 1;
		std::vector<uint8_t> bi;
		std::vector<vqvdb::Coord3i> bo;
		while (r.hasNext()) {
			const size_t n = r.nextBatch(97, bi, bo);
			if (std::memcmp(bi.data(), idx.data() + seen * 64, n * 64) != 0 || std::memcmp(bo.data(), org.data() + seen, n * 12) != 0) return 1;
			seen += n;
		}
		++grids;
	}
	std::printf("streamtest: %d grids, %zu leaves round-tripped\n", grids, seen);
	return (grids == 2 && seen == 1000) ? 0 : 1;
}

// vqvdb::StreamReader over tests/golden/ref_writer_v3.vqvdb — written by the reference's REAL VDBStreamWriter
// (src/Utils/VQVDB_Reader.cpp:58-150; generator tools/prove_vqvdb_format.py, which documents the content): 700 + 300 leaves
int readcheck(const std::string& path, size_t batch) {
	vqvdb::StreamReader r(path);
	const char* names[2] = {"density", "temperature"};
	const size_t counts[2] = {700, 300};
	size_t base = 0;
	int g = 0;
	while (r.hasNextGrid()) {
		const vqvdb::GridMeta m = r.nextGrid();
		if (g > 1 || m.name != names[g] || m.totalBlocks != counts[g] || m.numEmbeddings != 256 || m.latentShape != std::vector<int64_t>{4, 4, 4}) return 10 + g;
		if (m.transform[0] != 0.25f * (g + 1) || m.transform[13] != 3.0f * (g + 1) || m.transform[15] != 1.0f) return 20 + g;
		std::vector<uint8_t> bi;
		std::vector<vqvdb::Coord3i> bo;
		size_t seen = 0;
		while (r.hasNext()) {
			const size_t n = r.nextBatch(batch, bi, bo);
			for (size_t i = 0; i < n; ++i) {
				const size_t j = base + seen + i;
				if (bo[i].x != int32_t(j % 37) * 8 - 64 || bo[i].y != int32_t((j / 37) % 41) * 8 || bo[i].z != -int32_t(j / 1517) * 8) return 30 + g;
				for (size_t k = 0; k < 64; ++k)
					if (bi[i * 64 + k] != static_cast<uint8_t>(((j * 64 + k) * 2654435761u) >> 24)) return 40 + g;
			}
			seen += n;
		}
		if (seen != counts[g]) return 50 + g;
		base += seen;
		++g;
	}
	std::printf("readcheck: %d grids, %zu leaves, every field as the reference writer stored it\n", g, base);
	return g == 2 ? 0 : 60;
}

// The reference orchestrator's two serial loops on synthetic leaves, timed per phase (bench.py -> "orchestrator_loop"):
//   compress   (VQVAECodec.cpp:78-134):  per batch { fresh pack buffer + memcpy of the leaves | backend->encode | writeBatch }
//   decompress (VQVAECodec.cpp:137-208): per batch { nextBatch | backend->decode (fresh zero-filled Tensor) | per-leaf memcpy into a 2 KiB leaf buffer }
// One backend per batch size (the SOP node cache keeps it across cooks, SOP_VQVDB_Encoder.hpp:43-50); the first call of each
// direction (lazy device allocations) is reported separately and excluded from the per-call figures, not from the totals.
// A fixed set of worker threads that split [0, n) into equal ranges: what tbb::parallel_for gives the reference's pack loop and leaf
// copies (VQVAECodec.cpp:50,182) on hardware_concurrency() / 2 cores, without TBB.  Batches of up to 1024 leaves run on the caller.
class RangePool {
public:
	explicit RangePool(unsigned workers) {
		for (unsigned w = 0; w + 1 < workers; ++w) th_.emplace_back([this, w] { loop(w); });
	}
	~RangePool() {
		{ std::lock_guard<std::mutex> lk(mu_); quit_ = true; ++gen_; }
		cv_.notify_all();
		for (auto& t : th_) t.join();
	}
	unsigned size() const { return static_cast<unsigned>(th_.size()) + 1; }
	template <typename F> void run(size_t n, F&& f) {
		const size_t parts = std::min<size_t>(size(), (n + 1023) / 1024);   // (2 MB of leaves per range at least: waking the pool costs ~0.1 ms)
		if (parts <= 1) { f(size_t(0), n); return; }
		fn_ = [&](size_t a, size_t b) { f(a, b); };
		{ std::lock_guard<std::mutex> lk(mu_); n_ = n; parts_ = parts; left_ = parts - 1; ++gen_; }
		cv_.notify_all();
		f((parts - 1) * n / parts, n);            // the caller takes the last range
		std::unique_lock<std::mutex> lk(mu_);
		done_.wait(lk, [this] { return left_ == 0; });
	}
private:
	void loop(unsigned w) {
		uint64_t seen = 0;
		for (;;) {
			size_t n, parts;
			{
				std::unique_lock<std::mutex> lk(mu_);
				cv_.wait(lk, [&] { return gen_ != seen; });
				seen = gen_;
				if (quit_) return;
				n = n_, parts = parts_;
			}
			if (w + 1 < parts) {
				fn_(w * n / parts, (w + 1) * n / parts);
				std::lock_guard<std::mutex> lk(mu_);
				if (--left_ == 0) done_.notify_one();
			}
		}
	}
	std::vector<std::thread> th_;
	std::mutex mu_;
	std::condition_variable cv_, done_;
	std::function<void(size_t, size_t)> fn_;
	size_t n_ = 0, parts_ = 0, left_ = 0;
	uint64_t gen_ = 0;
	bool quit_ = false;
};

int loopbench(const std::string& pack, size_t total, const std::string& tmp, const std::string& batches, unsigned threads) {
	RangePool pool(std::max(1u, threads));
	const char* tag = threads > 1 ? "loopbench-mt" : "loopbench";
	constexpr size_t BASE = 65536;
	std::vector<float> base(std::min(total, BASE) * LEAF_VOXELS);
	uint32_t x = 2463534242u;
	for (float& v : base) { x ^= x << 13; x ^= x >> 17; x ^= x << 5; v = static_cast<float>(x >> 8) * (1.0f / 16777216.0f); }
	const size_t nbase = base.size() / LEAF_VOXELS;
	std::vector<std::unique_ptr<float[]>> leafStore;   // decompress side: 2 KiB per leaf, allocated up front (the tree owns its leaves)
	using clk = std::chrono::steady_clock;
	auto ms = [](clk::time_point a, clk::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
	size_t pos = 0;
	while (pos < batches.size()) {
		const size_t comma = batches.find(',', pos);
		const size_t batch = std::stoul(batches.substr(pos, comma == std::string::npos ? std::string::npos : comma - pos));
		pos = comma == std::string::npos ? batches.size() : comma + 1;
		auto backend = makeBackend(pack);
		double tPack = 0, tCall = 0, tWrite = 0, first = 0;
		size_t calls = 0;
		const auto c0 = clk::now();
		{
			vqvdb::StreamWriter writer(tmp);
			vqvdb::GridMeta meta;
			meta.name = "density";
			meta.latentShape = backend->getLatentShape();
			meta.totalBlocks = total;
			writer.startGrid(meta);
			for (size_t start = 0; start < total; start += batch, ++calls) {
				const size_t B = std::min(batch, total - start);
				const auto t0 = clk::now();
				std::vector<float> hostData(B * LEAF_VOXELS);
				std::vector<vqvdb::Coord3i> origins(B);
				pool.run(B, [&](size_t lo, size_t hi) {
					for (size_t i = lo; i < hi; ++i) {
						origins[i] = originOf(start + i);
						std::memcpy(hostData.data() + i * LEAF_VOXELS, base.data() + ((start + i) % nbase) * LEAF_VOXELS, LEAF_VOXELS * sizeof(float));
					}
				});
				TensorView view;
				view.data = hostData.data();
				view.shape = {static_cast<int64_t>(B), 1, 8, 8, 8};
				view.dtype = DataType::FLOAT32;
				const auto t1 = clk::now();
				const Tensor encoded = backend->encode(view);
				const auto t2 = clk::now();
				writer.writeBatch(encoded.getData<uint8_t>(), origins.data(), B);
				const auto t3 = clk::now();
				if (calls == 0) first = ms(t1, t2);
				else tPack += ms(t0, t1), tCall += ms(t1, t2), tWrite += ms(t2, t3);
			}
			writer.endGrid();
			writer.close();
		}
		const double wallC = ms(c0, clk::now());
		const double nc = calls > 1 ? double(calls - 1) : 1.0;
		std::printf("%s compress   batch %zu: %zu leaves in %.1f ms = %.4f M leaves/s | first call %.3f ms | per call: pack %.4f ms, encode %.4f ms, frame+write %.4f ms\n",
		            tag, batch, total, wallC, total / wallC / 1e3, first, tPack / nc, tCall / nc, tWrite / nc);
		if (leafStore.empty()) {
			leafStore.resize((total + BASE - 1) / BASE);
			for (size_t i = 0; i < leafStore.size(); ++i) { leafStore[i].reset(new float[BASE * LEAF_VOXELS]); std::memset(leafStore[i].get(), 0, BASE * LEAF_VOXELS * sizeof(float)); }
		}
		double tRead = 0, tDec = 0, tCopy = 0;
		first = 0, calls = 0;
		size_t leafNo = 0;
		const auto d0 = clk::now();
		{
			vqvdb::StreamReader reader(tmp);
			while (reader.hasNextGrid()) {
				const vqvdb::GridMeta meta = reader.nextGrid();
				std::vector<uint8_t> idx;
				std::vector<vqvdb::Coord3i> origins;
				while (reader.hasNext()) {
					const auto t0 = clk::now();
					const size_t B = reader.nextBatch(batch, idx, origins);
					if (B == 0) break;
					TensorView view;
					view.data = idx.data();
					view.shape = {static_cast<int64_t>(B)};
					view.shape.insert(view.shape.end(), meta.latentShape.begin(), meta.latentShape.end());
					view.dtype = DataType::UINT8;
					const auto t1 = clk::now();
					const Tensor decoded = backend->decode(view);
					const auto t2 = clk::now();
					const float* src = decoded.getData<float>();
					pool.run(B, [&](size_t lo, size_t hi) {
						for (size_t i = lo; i < hi; ++i) {  // stand-in for touchLeaf + memcpy + setValuesOn (VQVAECodec.cpp:182-192)
							const size_t j = leafNo + i;
							std::memcpy(leafStore[j / BASE].get() + (j % BASE) * LEAF_VOXELS, src + i * LEAF_VOXELS, LEAF_VOXELS * sizeof(float));
						}
					});
					const auto t3 = clk::now();
					if (calls == 0) first = ms(t1, t2);
					else tRead += ms(t0, t1), tDec += ms(t1, t2), tCopy += ms(t2, t3);
					leafNo += B;
					++calls;
				}
			}
		}
		const double wallD = ms(d0, clk::now());
		const double nd = calls > 1 ? double(calls - 1) : 1.0;
		std::printf("%s decompress batch %zu: %zu leaves in %.1f ms = %.4f M leaves/s | first call %.3f ms | per call: read+deframe %.4f ms, decode %.4f ms, leaf copies %.4f ms\n",
		            tag,
		            batch, leafNo, wallD, leafNo / wallD / 1e3, first, tRead / nd, tDec / nd, tCopy / nd);
		if (leafNo != total) return 1;
	}
	std::remove(tmp.c_str());
	return 0;
}

// The same two loops with the leaf-pointer entry points of the C ABI: no pack buffer, no owning Tensor,
// no per-leaf copies on the caller's side — the loop hands over leaf.buffer().data() pointers (VQVAECodec.cpp:36-59,182-192 replaced).
// Leaves live the way an OpenVDB tree keeps them: one heap block of 2 KiB per leaf, visited in an order unrelated to the allocation order.
int loopbenchPtrs(const std::string& pack, size_t total, const std::string& tmp, const std::string& batches) {
	constexpr size_t BASE = 65536;
	std::vector<float> base(std::min(total, BASE) * LEAF_VOXELS);
	uint32_t x = 2463534242u;
	for (float& v : base) { x ^= x << 13; x ^= x >> 17; x ^= x << 5; v = static_cast<float>(x >> 8) * (1.0f / 16777216.0f); }
	const size_t nbase = base.size() / LEAF_VOXELS;
	std::vector<std::unique_ptr<float[]>> blocks(total);
	for (auto& b : blocks) b.reset(new float[LEAF_VOXELS]);
	std::vector<float*> leafOf(total);
	for (size_t i = 0; i < total; ++i) leafOf[i] = blocks[i].get();
	for (size_t i = total; i > 1; --i) { x ^= x << 13; x ^= x >> 17; x ^= x << 5; std::swap(leafOf[i - 1], leafOf[x % i]); }   // shuffled: leaf i sits in an arbitrary block
	for (size_t i = 0; i < total; ++i) std::memcpy(leafOf[i], base.data() + (i % nbase) * LEAF_VOXELS, LEAF_VOXELS * sizeof(float));
	using clk = std::chrono::steady_clock;
	auto ms = [](clk::time_point a, clk::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
	size_t pos = 0;
	while (pos < batches.size()) {
		const size_t comma = batches.find(',', pos);
		const size_t batch = std::stoul(batches.substr(pos, comma == std::string::npos ? std::string::npos : comma - pos));
		pos = comma == std::string::npos ? batches.size() : comma + 1;
		auto backend = makeBackend(pack);
		vqhip_codec* h = static_cast<HipBackend*>(backend.get())->handle();
		auto check = [&](int rc) { if (rc != VQHIP_OK) throw std::runtime_error(vqhip_last_error(h)); };
		double tPtr = 0, tCall = 0, tWrite = 0, first = 0;
		size_t calls = 0;
		std::vector<const float*> src(batch);
		std::vector<uint8_t> idx(batch * 64);
		std::vector<vqvdb::Coord3i> origins(batch);
		const auto c0 = clk::now();
		{
			vqvdb::StreamWriter writer(tmp);
			vqvdb::GridMeta meta;
			meta.name = "density";
			meta.latentShape = backend->getLatentShape();
			meta.totalBlocks = total;
			writer.startGrid(meta);
			for (size_t start = 0; start < total; start += batch, ++calls) {
				const size_t B = std::min(batch, total - start);
				const auto t0 = clk::now();
				for (size_t i = 0; i < B; ++i) origins[i] = originOf(start + i), src[i] = leafOf[start + i];
				const auto t1 = clk::now();
				check(vqhip_encode_leaves(h, src.data(), static_cast<int64_t>(B), idx.data()));
				const auto t2 = clk::now();
				writer.writeBatch(idx.data(), origins.data(), B);
				const auto t3 = clk::now();
				if (calls == 0) first = ms(t1, t2);
				else tPtr += ms(t0, t1), tCall += ms(t1, t2), tWrite += ms(t2, t3);
			}
			writer.endGrid();
			writer.close();
		}
		const double wallC = ms(c0, clk::now());
		const double nc = calls > 1 ? double(calls - 1) : 1.0;
		std::printf("loopbench-ptrs compress   batch %zu: %zu leaves in %.1f ms = %.4f M leaves/s | first call %.3f ms | per call: pointers %.4f ms, encode %.4f ms, frame+write %.4f ms\n",
		            batch, total, wallC, total / wallC / 1e3, first, tPtr / nc, tCall / nc, tWrite / nc);
		for (size_t i = 0; i < total; i += 4097) leafOf[i][0] = -1.0f;   // (decode must overwrite these)
		double tRead = 0, tDec = 0;
		first = 0, calls = 0;
		size_t leafNo = 0;
		std::vector<float*> dst(batch);
		const auto d0 = clk::now();
		{
			vqvdb::StreamReader reader(tmp);
			while (reader.hasNextGrid()) {
				reader.nextGrid();
				std::vector<uint8_t> ridx;
				std::vector<vqvdb::Coord3i> rorg;
				while (reader.hasNext()) {
					const auto t0 = clk::now();
					const size_t B = reader.nextBatch(batch, ridx, rorg);
					if (B == 0) break;
					for (size_t i = 0; i < B; ++i) dst[i] = leafOf[leafNo + i];   // stand-in for touchLeaf(origin)->buffer().data()
					const auto t1 = clk::now();
					check(vqhip_decode_leaves(h, ridx.data(), static_cast<int64_t>(B), dst.data()));
					const auto t2 = clk::now();
					if (calls == 0) first = ms(t1, t2);
					else tRead += ms(t0, t1), tDec += ms(t1, t2);
					leafNo += B;
					++calls;
				}
			}
		}
		const double wallD = ms(d0, clk::now());
		const double nd = calls > 1 ? double(calls - 1) : 1.0;
		std::printf("loopbench-ptrs decompress batch %zu: %zu leaves in %.1f ms = %.4f M leaves/s | first call %.3f ms | per call: read+deframe+pointers %.4f ms, decode %.4f ms, leaf copies %.4f ms\n",
		            batch, leafNo, wallD, leafNo / wallD / 1e3, first, tRead / nd, tDec / nd, 0.0);
		if (leafNo != total) return 1;
		for (size_t i = 0; i < total; i += 4097)
			if (!(leafOf[i][0] > 0.0f && leafOf[i][0] < 1.0f)) { std::fprintf(stderr, "loopbench-ptrs: leaf %zu was not written\n", i); return 1; }
		for (size_t i = 0; i < total; ++i) std::memcpy(leafOf[i], base.data() + (i % nbase) * LEAF_VOXELS, LEAF_VOXELS * sizeof(float));   // the next batch size compresses the same input
	}
	std::remove(tmp.c_str());
	return 0;
}

int errors(const std::string& pack) {
	int bad = 0;
	auto expectThrow = [&](const char* what, auto&& fn, const char* msg) {
		try {
			fn();
			std::printf("FAIL %s: no exception\n", what);
			++bad;
		} catch (const std::runtime_error& e) {
			if (std::string(e.what()).find(msg) == std::string::npos) { std::printf("FAIL %s: '%s'\n", what, e.what()); ++bad; }
		}
	};
	auto backend = makeBackend(pack);
	std::vector<float> leaf(LEAF_VOXELS, 0.25f);
	TensorView v;
	v.data = leaf.data();
	v.shape = {1, 1, 8, 8, 8};
	v.dtype = DataType::UINT8;
	expectThrow("encode dtype", [&] { backend->encode(v); }, "encode expects FLOAT32 data.");
	v.dtype = DataType::FLOAT32;
	expectThrow("decode dtype", [&] { backend->decode(v); }, "decode expects UINT8 data.");
	v.shape = {1, 8, 8, 8};
	expectThrow("encode shape", [&] { backend->encode(v); }, "encode expects shape");
	CodecConfig cfg;
	cfg.device = CodecConfig::Device::CUDA;
	cfg.source = std::filesystem::path("/nonexistent/model.vqw");
	if (IVQVAECodec::create(cfg, BackendType::HIP) != nullptr) { std::printf("FAIL create(missing pack) != nullptr\n"); ++bad; }
	cfg.source = std::filesystem::path(pack);
	if (IVQVAECodec::create(cfg, BackendType::ONNX) != nullptr) { std::printf("FAIL create(ONNX) != nullptr\n"); ++bad; }
	cfg.device = CodecConfig::Device::CPU;
	if (IVQVAECodec::create(cfg, BackendType::HIP) != nullptr) { std::printf("FAIL create(CPU) != nullptr\n"); ++bad; }
	std::printf(bad ? "errors: %d failures\n" : "errors: all behaviours match (%d failures)\n", bad);
	return bad ? 1 : 0;
}

// Two codecs, two caller threads — the SOP node caches own one codec each (SOP_VQVDB_Encoder.hpp:43-50) and Houdini may cook an encoder
// node and a decoder node at the same time.  Thread A encodes on backend 1 while thread B decodes on backend 2 (mixed batch sizes), then
// B creates / destroys backends while A keeps encoding; every result must equal the serial run's bytes.
int threads(const std::string& pack, size_t iters) {
	auto be1 = makeBackend(pack), be2 = makeBackend(pack);
	const size_t sizes[] = {64, 1, 333, 1024, 4096, 20000, 97, 8192};
	constexpr size_t NS = sizeof(sizes) / sizeof(sizes[0]), MAXB = 20000;
	std::vector<float> leaves(MAXB * LEAF_VOXELS);
	uint32_t st = 12345u;
	for (auto& v : leaves) { st = st * 1664525u + 1013904223u; v = static_cast<float>(st >> 8) * (1.0f / 16777216.0f); }
	auto view = [&](const void* p, size_t B, bool enc) {
		TensorView v;
		v.data = p;
		v.shape = enc ? std::vector<int64_t>{static_cast<int64_t>(B), 1, 8, 8, 8} : std::vector<int64_t>{static_cast<int64_t>(B), 4, 4, 4};
		v.dtype = enc ? DataType::FLOAT32 : DataType::UINT8;
		return v;
	};
	// serial truth on a third backend
	std::vector<std::vector<std::byte>> wantIdx(NS), wantRec(NS);
	{
		auto be0 = makeBackend(pack);
		for (size_t k = 0; k < NS; ++k) {
			const Tensor e = be0->encode(view(leaves.data(), sizes[k], true));
			wantIdx[k] = e.buffer;
			wantRec[k] = be0->decode(view(wantIdx[k].data(), sizes[k], false)).buffer;
		}
	}
	std::atomic<int> bad{0};
	std::atomic<bool> stop{false};
	std::string firstError;
	std::mutex emu;
	auto guard = [&](auto&& fn) {
		try { fn(); } catch (const std::exception& e) { std::lock_guard<std::mutex> lk(emu); if (firstError.empty()) firstError = e.what(); ++bad; }
	};
	{	// phase 1: encode on backend 1 || decode on backend 2
		std::thread A([&] { guard([&] { for (size_t i = 0; i < iters; ++i) { const size_t k = i % NS; if (be1->encode(view(leaves.data(), sizes[k], true)).buffer != wantIdx[k]) ++bad; } }); });
		std::thread B([&] { guard([&] { for (size_t i = 0; i < iters; ++i) { const size_t k = (i * 3 + 1) % NS; if (be2->decode(view(wantIdx[k].data(), sizes[k], false)).buffer != wantRec[k]) ++bad; } }); });
		A.join();
		B.join();
	}
	size_t churn = 0, encodes = 0;
	{	// phase 2: create / destroy churn on B while A keeps encoding
		std::thread A([&] { guard([&] { for (size_t i = 0; !stop.load(); ++i, ++encodes) { const size_t k = i % NS; if (be1->encode(view(leaves.data(), sizes[k], true)).buffer != wantIdx[k]) ++bad; } }); });
		std::thread B([&] {
			guard([&] {
				for (size_t i = 0; i < std::max<size_t>(iters / 4, 8); ++i, ++churn) {
					auto tmp = makeBackend(pack);
					const size_t k = i % 3;
					if (tmp->decode(view(wantIdx[k].data(), sizes[k], false)).buffer != wantRec[k]) ++bad;
				}
			});
			stop = true;
		});
		A.join();
		B.join();
	}
	std::printf("threads: %zu iterations encode || decode on two backends, %zu create/destroy cycles beside %zu encodes: %d mismatches%s%s\n", iters, churn, encodes,
	            bad.load(), firstError.empty() ? "" : ", first exception: ", firstError.c_str());
	return bad ? 1 : 0;
}
}  // namespace

int main(int argc, char** argv) {
	try {
		const std::string mode = argc > 1 ? argv[1] : "";
		if (mode == "compress" && argc == 6) return compress(argv[2], argv[3], argv[4], std::stoul(argv[5]));
		if (mode == "decompress" && argc == 6) return decompress(argv[2], argv[3], argv[4], std::stoul(argv[5]));
		if (mode == "compress_stream" && argc == 6) return compressStream(argv[2], argv[3], argv[4], std::stoul(argv[5]));
		if (mode == "decompress_stream" && argc == 6) return decompressStream(argv[2], argv[3], argv[4], std::stoul(argv[5]));
		if (mode == "errors" && argc == 3) return errors(argv[2]);
		if (mode == "threads" && argc == 4) return threads(argv[2], std::stoul(argv[3]));
		if (mode == "loopbench" && argc == 6) return loopbench(argv[2], std::stoul(argv[3]), argv[4], argv[5], 1);
		// ... with the pack loop and the leaf copies on N threads (0 = hardware_concurrency() / 2: what tbb::parallel_for uses in the reference)
		if (mode == "loopbench" && argc == 7) {
			const unsigned t = static_cast<unsigned>(std::stoul(argv[6]));
			return loopbench(argv[2], std::stoul(argv[3]), argv[4], argv[5], t ? t : std::max(1u, std::thread::hardware_concurrency() / 2));
		}
		if (mode == "loopbench_ptrs" && argc == 6) return loopbenchPtrs(argv[2], std::stoul(argv[3]), argv[4], argv[5]);
		if (mode == "streamtest" && argc == 3) return streamtest(argv[2]);
		if (mode == "readcheck" && argc == 4) return readcheck(argv[2], std::stoul(argv[3]));
		if (mode == "makefile" && argc == 4) return makefile(argv[2], std::stoul(argv[3]));
		std::fprintf(stderr, "usage: leaf_harness compress|decompress|errors ...\n");
		return 2;
	} catch (const std::exception& e) {
		std::fprintf(stderr, "leaf_harness: %s\n", e.what());
		return 1;
	}
}
