// TSAN driver for the host-only parts of libvqvdb_hip (no GPU in the build container): concurrent creates that fail at every stage
// (missing file, bad magic, truncated table, valid pack -> no device), thread-local error strings, the multi-device front end's error path.
#include <atomic>
#include <cstdio>
#include <cstring>
#include <string>
#include <thread>
#include <vector>
#include "vqvdb_hip.h"
int main(int argc, char** argv) {
	std::vector<unsigned char> pack;
	if (argc > 1) { FILE* f = std::fopen(argv[1], "rb"); if (f) { unsigned char b[65536]; size_t n; while ((n = std::fread(b, 1, sizeof b, f)) > 0) pack.insert(pack.end(), b, b + n); std::fclose(f); } }
	std::atomic<int> bad{0};
	auto worker = [&](int t) {
		for (int i = 0; i < 200; ++i) {
			vqhip_codec* h = nullptr;
			int rc;
			std::string want;
			switch ((i + t) % 5) {
				case 0: rc = vqhip_create("/nonexistent/m.vqw", nullptr, 0, 0, &h); want = "Model file not found"; break;
				case 1: { unsigned char junk[64]; std::memset(junk, 'x', 64); rc = vqhip_create(nullptr, junk, 64, 0, &h); want = "bad magic"; break; }
				case 2: rc = vqhip_create(nullptr, pack.data(), pack.size() / 3, 0, &h); want = ""; break;   // truncated
				case 3: rc = vqhip_create(nullptr, pack.data(), pack.size(), 0, &h); want = "no HIP device"; break;
				default: { int devs[3] = {0, 1, 2}; vqhip_multi* m = nullptr; rc = vqhip_multi_create(nullptr, pack.data(), pack.size(), devs, 3, &m); want = ""; if (m) vqhip_multi_destroy(m); break; }
			}
			const std::string msg = vqhip_last_error(nullptr);
			if (rc == VQHIP_OK || h || msg.empty() || (!want.empty() && msg.find(want) == std::string::npos)) { ++bad; std::fprintf(stderr, "thread %d iter %d: rc %d '%s' (wanted '%s')\n", t, i, rc, msg.c_str(), want.c_str()); }
		}
	};
	std::vector<std::thread> th;
	for (int t = 0; t < 8; ++t) th.emplace_back(worker, t);
	for (auto& x : th) x.join();
	std::printf("tsan host driver: 8 threads x 200 failing creates, %d unexpected results, version %s\n", bad.load(), vqhip_version());
	return bad ? 1 : 0;
}
