// vision.hpp — C++ host-side mirrors of lmrs::vision::VisionTransformer and lmrs::processor::PHI3VProcessor over the C ABI
// (reference src/vision.rs:99-243 new, :244-577 forward; src/processor.rs:168-232 new, :234-342 forward).  Same names, argument
// meaning and tuple returns as the reference; its panics are exceptions here.  PHI3VProcessor::process (image resize / pad /
// normalise, processor.rs:344-375) is host image code outside the device path and is not mirrored.  Header-only.
#pragma once
#include <cstddef>
#include <cstdint>
#include <utility>
#include <vector>

#include "transformer.hpp"

namespace lmrs_host {

class VisionTransformer {
public:
    // VisionTransformer::new(data) -> (VisionTransformer, usize): `data` starts at the offset Transformer::new returned.
    static std::pair<VisionTransformer, std::size_t> create(const std::uint8_t* data, std::size_t len, int device = 0) {
        VisionTransformer v; std::size_t used = 0;
        check(lmrs_vision_create(data, len, device, &v.h_, &used));
        return {std::move(v), used};
    }
    VisionTransformer(VisionTransformer&& o) noexcept : h_(o.h_) { o.h_ = nullptr; }
    VisionTransformer(const VisionTransformer&) = delete;
    VisionTransformer& operator=(const VisionTransformer&) = delete;
    ~VisionTransformer() { if (h_) lmrs_vision_destroy(h_); }

    // forward(&mut self, pixel_values, num_crops) -> (Vec<f32>, u32): patch features without the class token, and their
    // per-crop length (576 * dim).  pixel_values: num_crops x 576 patches x 588 floats, as process() lays them out.
    std::pair<std::vector<float>, std::uint32_t> forward(const std::vector<float>& pixel_values, std::uint32_t num_crops) {
        if (pixel_values.size() != static_cast<std::size_t>(num_crops) * 576 * 588) throw Panic("pixel_values: num_crops * 576 * 588 floats expected");
        std::vector<float> out(static_cast<std::size_t>(num_crops) * 576 * 1024);
        std::uint32_t new_shape = 0;
        check(lmrs_vision_forward(h_, pixel_values.data(), num_crops, out.data(), &new_shape));
        out.resize(static_cast<std::size_t>(num_crops) * new_shape);
        return {std::move(out), new_shape};
    }

private:
    VisionTransformer() = default;
    lmrs_vision* h_ = nullptr;
};

class PHI3VProcessor {
public:
    // PHI3VProcessor::new(data) -> PHI3VProcessor: `data` starts where the vision section ended.
    static PHI3VProcessor create(const std::uint8_t* data, std::size_t len, int device = 0) {
        PHI3VProcessor p;
        check(lmrs_processor_create(data, len, device, &p.h_, nullptr));
        p.text_dim_ = reinterpret_cast<const std::uint32_t*>(data)[1];
        return p;
    }
    PHI3VProcessor(PHI3VProcessor&& o) noexcept : h_(o.h_), text_dim_(o.text_dim_) { o.h_ = nullptr; }
    PHI3VProcessor(const PHI3VProcessor&) = delete;
    PHI3VProcessor& operator=(const PHI3VProcessor&) = delete;
    ~PHI3VProcessor() { if (h_) lmrs_processor_destroy(h_); }

    // forward(&self, out_patches, new_shape, patch_side, w_crop, h_crop) -> Vec<f32>: num_embeds x text_dim image embeddings,
    // num_embeds = (h_crop*patch_side) * (w_crop*patch_side + 1) + patch_side * (patch_side + 1) + 1.
    std::vector<float> forward(const std::vector<float>& out_patches, std::uint32_t new_shape, std::uint32_t patch_side, std::uint32_t w_crop, std::uint32_t h_crop) {
        const std::size_t ne = static_cast<std::size_t>(h_crop * patch_side) * (w_crop * patch_side + 1) + static_cast<std::size_t>(patch_side) * (patch_side + 1) + 1;
        std::vector<float> out(ne * text_dim_);
        std::uint32_t n = 0;
        check(lmrs_processor_forward(h_, out_patches.data(), static_cast<std::uint32_t>(out_patches.size()), new_shape, patch_side, w_crop, h_crop, out.data(), &n));
        out.resize(static_cast<std::size_t>(n) * text_dim_);
        return out;
    }

private:
    PHI3VProcessor() = default;
    lmrs_processor* h_ = nullptr;
    std::uint32_t text_dim_ = 0;
};

}  // namespace lmrs_host
