// vqvdb_hip_backend.hpp — `HipBackend final : IVQVAECodec`, the MI355X sibling of the
// reference's TorchBackend (src/backends/torch/TorchBackend.{hpp,cpp}) and OnnxCudaBackend
// (src/backends/onnx/OnnxBackend_Cuda.cpp).  Header-only glue between the reference's C++
// plugin surface and the C ABI of libvqvdb_hip.so (include/vqvdb_hip.h): TensorView/Tensor
// <-> raw pointers, status codes <-> std::runtime_error.  No tensor arithmetic lives here.
//
// In the reference tree:   #include "core/IVQVAECodec.hpp" is found on the include path.
// Standalone (this repo):  define VQVDB_HIP_STANDALONE to use vqvdb_amd/host/codec_interface.hpp.
#pragma once
#ifdef VQVDB_HIP_STANDALONE
#include "../vqvdb_amd/host/codec_interface.hpp"
#else
#include "core/IVQVAECodec.hpp"
#endif

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <stdexcept>
#include <string>

#include "vqvdb_hip.h"

class HipBackend final : public IVQVAECodec {
   public:
	friend std::unique_ptr<IVQVAECodec> IVQVAECodec::create(const CodecConfig& config, BackendType type);
	~HipBackend() override { vqhip_destroy(codec_); }
	HipBackend(const HipBackend&) = delete;
	HipBackend& operator=(const HipBackend&) = delete;

	// Same guards and messages as TorchBackend::encode (TorchBackend.cpp:133-136).
	Tensor encode(const TensorView& leafBatch) const override {
		if (leafBatch.dtype != DataType::FLOAT32) throw std::runtime_error("encode expects FLOAT32 data.");
		const int64_t B = checkedBatch(leafBatch, {1, 8, 8, 8}, "encode expects shape [B,1,8,8,8].");
		Tensor result;
		result.shape = {B, latentShape_[0], latentShape_[1], latentShape_[2]};
		result.dtype = DataType::UINT8;
		result.buffer.resize(static_cast<size_t>(B) * VQHIP_LATENT_VOXELS);
		check(vqhip_encode(codec_, static_cast<const float*>(leafBatch.data), B, reinterpret_cast<uint8_t*>(result.buffer.data())));
		return result;
	}

	// TorchBackend::decode (TorchBackend.cpp:166-194): UINT8 [B,4,4,4] in, FLOAT32 [B,1,8,8,8] out (5-D).
	Tensor decode(const TensorView& indices) const override {
		if (indices.dtype != DataType::UINT8) throw std::runtime_error("decode expects UINT8 data.");
		const int64_t B = checkedBatch(indices, latentShape_, "decode expects shape [B,4,4,4].");
		Tensor result;
		result.shape = {B, 1, 8, 8, 8};
		result.dtype = DataType::FLOAT32;
		result.buffer.resize(static_cast<size_t>(B) * VQHIP_LEAF_VOXELS * sizeof(float));
		check(vqhip_decode(codec_, static_cast<const uint8_t*>(indices.data), B, reinterpret_cast<float*>(result.buffer.data())));
		return result;
	}

	const std::vector<int64_t>& getLatentShape() const override { return latentShape_; }

	// Extension: the C handle, for callers that use the leaf-pointer or whole-file entry points of vqvdb_hip.h
	// (vqhip_encode_leaves, vqhip_decompress_file, vqhip_reserve ...) next to the IVQVAECodec interface.
	vqhip_codec* handle() const { return codec_; }

   private:
	explicit HipBackend(const CodecConfig& config) {
		if (config.device != CodecConfig::Device::CUDA)
			throw std::runtime_error("HIP backend requires Device::CUDA (GPU); it has no CPU path.");
		int rc;
		if (std::holds_alternative<std::filesystem::path>(config.source)) {
			rc = vqhip_create(std::get<std::filesystem::path>(config.source).string().c_str(), nullptr, 0, deviceId(), &codec_);
		} else if (std::holds_alternative<EmbeddedModel>(config.source)) {
			// embedded weight pack: see the build modes at the end of this header
			rc = vqhip_create(nullptr, embeddedData(), embeddedSize(), deviceId(), &codec_);
		} else {
			throw std::logic_error("Unsupported model source type.");
		}
		if (rc != VQHIP_OK) throw std::runtime_error(vqhip_last_error(nullptr));
		int64_t ls[3];
		check(vqhip_latent_shape(codec_, ls));
		latentShape_.assign(ls, ls + 3);
	}
	static int deviceId() {
		const char* e = std::getenv("VQVDB_HIP_DEVICE");
		return e ? std::atoi(e) : 0;  // the reference hard-codes device 0 (OnnxBackend_Cuda.cpp:21)
	}
	static const void* embeddedData();
	static size_t embeddedSize();
	void check(int rc) const {
		if (rc != VQHIP_OK) throw std::runtime_error(vqhip_last_error(codec_));
	}
	static int64_t checkedBatch(const TensorView& v, const std::vector<int64_t>& tail, const char* msg) {
		if (v.shape.size() != tail.size() + 1 || v.shape[0] < 1 || !std::equal(tail.begin(), tail.end(), v.shape.begin() + 1) || !v.data)
			throw std::runtime_error(msg);
		return v.shape[0];
	}
	vqhip_codec* codec_ = nullptr;
	std::vector<int64_t> latentShape_;
};

// Embedded weight pack — CodecConfig::source = EmbeddedModel{}, the value both SOPs hard-code (SOP_VQVDB_Encoder.cpp:63-67,
// SOP_VQVDB_Decoder.cpp:58-62).  The role bin/bin_model.h plays for TorchBackend.cpp:20,39-43.  Three build modes:
//   -DVQVDB_HIP_EMBEDDED_PACK_HEADER='"bin/vqhip_pack.h"'   the adapter includes the generated header itself (one TU: the factory).
//        Works with `python -m vqvdb_amd.weightpack --header m.vqw vqhip_pack.h` AND with the reference's own
//        `python/convert_to_header.py m.vqw vqhip_pack.h --name g_vqhip_pack_data` (whose size object is called
//        g_vqhip_pack_data_size: the size is taken from the array type, which both tools complete).
//   -DVQVDB_HIP_EMBEDDED_PACK                                the two objects are linked in from a TU of their own
//        (`gcc -x c -c vqhip_pack.h`, output of vqvdb_amd.weightpack --header: external linkage, extern "C").
//   neither                                                   EmbeddedModel is refused with a message (no weights in this build).
#if defined(VQVDB_HIP_EMBEDDED_PACK_HEADER)
#include VQVDB_HIP_EMBEDDED_PACK_HEADER
inline const void* HipBackend::embeddedData() { return g_vqhip_pack_data; }
inline size_t HipBackend::embeddedSize() { return sizeof(g_vqhip_pack_data); }
#elif defined(VQVDB_HIP_EMBEDDED_PACK)
extern "C" {
extern const unsigned char g_vqhip_pack_data[];
extern const size_t g_vqhip_pack_size;
}
inline const void* HipBackend::embeddedData() { return g_vqhip_pack_data; }
inline size_t HipBackend::embeddedSize() { return g_vqhip_pack_size; }
#else
inline const void* HipBackend::embeddedData() { return nullptr; }
inline size_t HipBackend::embeddedSize() { return 0; }
#endif
