// codec_factory.cpp — IVQVAECodec::create for the standalone build: the reference factory's
// switch (src/core/IVQVAECodec.cpp:76-110) with the one new case.  Inside the reference tree
// the maintainer adds the same case under #ifdef ENABLE_HIP_BACKEND.
#define VQVDB_HIP_STANDALONE
#include <iostream>

#include "../../include/vqvdb_hip_backend.hpp"

std::unique_ptr<IVQVAECodec> IVQVAECodec::create(const CodecConfig& config, BackendType type) {
	try {
		switch (type) {
			case BackendType::HIP:
				return std::unique_ptr<IVQVAECodec>(new HipBackend(config));
			default:
				throw std::runtime_error("Requested backend type is not available or disabled in the build configuration.");
		}
	} catch (const std::exception& e) {
		std::cerr << "Failed to create VQ-VAE backend: " << e.what() << std::endl;
		return nullptr;
	}
}
