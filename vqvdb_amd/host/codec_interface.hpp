// codec_interface.hpp — standalone mirror of the reference's plugin surface for this path
// (src/core/IVQVAECodec.hpp:21-136): same type names, members and meaning, so that code written
// against the reference header compiles against this one unchanged.  Used ONLY when building
// outside the reference tree (tests, harness, bench); inside the reference tree the adapter
// include/vqvdb_hip_backend.hpp includes the reference's own "core/IVQVAECodec.hpp".
//
// The declarations restate the interface of ZephirFXEC/VQVDB's src/core/IVQVAECodec.hpp (BSD-3-Clause,
// Copyright (c) the VQVDB authors; see the upstream repository's LICENSE), which a drop-in backend has to
// match name for name.
//
// The one addition is BackendType::HIP, appended so the existing enumerator values keep
// their numbers.
#pragma once
#include <cstddef>
#include <cstdint>
#include <filesystem>
#include <memory>
#include <variant>
#include <vector>

enum class BackendType { LibTorch, ONNX, HIP };

struct EmbeddedModel {};
struct OnnxModelPaths {
	std::filesystem::path encoder_path;
	std::filesystem::path decoder_path;
};
using ModelSource = std::variant<EmbeddedModel, std::filesystem::path, OnnxModelPaths>;

enum class DataType { FLOAT32, UINT8 };

struct TensorView {  // non-owning, host memory, caller keeps ownership
	const void* data = nullptr;
	std::vector<int64_t> shape;
	DataType dtype;
};

struct Tensor {  // owning result
	std::vector<std::byte> buffer;
	std::vector<int64_t> shape;
	DataType dtype;
	template <typename T> const T* getData() const { return reinterpret_cast<const T*>(buffer.data()); }
	template <typename T> T* getData() { return reinterpret_cast<T*>(buffer.data()); }
};

struct CodecConfig {
	enum class Device { CPU, CUDA };  // CUDA is read as "GPU" by the HIP backend
	Device device = Device::CPU;
	ModelSource source = EmbeddedModel{};
};

class IVQVAECodec {
   public:
	virtual ~IVQVAECodec() = default;
	// Any failure is reported on std::cerr and yields nullptr (reference: IVQVAECodec.cpp:106-109).
	static std::unique_ptr<IVQVAECodec> create(const CodecConfig& config, BackendType type);
	virtual Tensor encode(const TensorView& leafBatch) const = 0;  // [B,1,8,8,8] FLOAT32 -> [B,4,4,4] UINT8
	virtual Tensor decode(const TensorView& indices) const = 0;    // [B,4,4,4] UINT8 -> [B,1,8,8,8] FLOAT32
	virtual const std::vector<int64_t>& getLatentShape() const = 0;
};
