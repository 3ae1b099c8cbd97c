// image_prefill.cpp — the multimodal prefill of the reference's chat loop (src/bin/chat.rs:84-121) over the C++ mirrors:
// Transformer::new -> VisionTransformer::new (at the offset it returned) -> PHI3VProcessor::new (after the vision section) ->
// vision.forward -> processor.forward -> the image features spliced between two get_embeddings blocks -> fill_kv_cache, then
// greedy decoding on the prefilled cache.  PHI3VProcessor::process (resize / pad / normalise / patchify, processor.rs:344-375)
// is host image code outside this library: the input here is its output, a raw f32 file of num_crops x 576 x 588 normalised
// patch values; the text prompt that follows the image is a list of token ids (tokenizer out of scope).
//   usage: image_prefill model.lmrs patches.f32 num_crops w_crop h_crop n_new token_id...
//   g++ -O2 -std=c++17 image_prefill.cpp -L.. -llmrs_hip -Wl,-rpath,'$ORIGIN/..' -o image_prefill
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <cstdio>
#include <cstdlib>
#include <fstream>

#include "vision.hpp"

int main(int argc, char** argv) {
    if (argc < 8) { std::fprintf(stderr, "usage: %s model.lmrs patches.f32 num_crops w_crop h_crop n_new token_id...\n", argv[0]); return 2; }
    const int fd = open(argv[1], O_RDONLY);
    if (fd < 0) { std::perror("open"); return 1; }
    struct stat st; fstat(fd, &st);
    const auto* data = static_cast<const std::uint8_t*>(mmap(nullptr, st.st_size, PROT_READ, MAP_PRIVATE, fd, 0));
    if (data == MAP_FAILED) { std::perror("mmap"); return 1; }
    const std::uint32_t num_crops = std::atoi(argv[3]), w_crop = std::atoi(argv[4]), h_crop = std::atoi(argv[5]), n_new = std::atoi(argv[6]);
    try {
        using namespace lmrs_host;
        auto [model, off_t] = Transformer::create(data, st.st_size);                                        // chat.rs:65
        if (!model.args.multimodal) throw Panic("Cannot use images in a non-multimodal model.");              // :85-88
        auto [vision, off_v] = VisionTransformer::create(data + off_t, st.st_size - off_t);                   // :90
        auto processor = PHI3VProcessor::create(data + off_t + off_v, st.st_size - off_t - off_v);            // :91
        std::vector<float> patches(static_cast<std::size_t>(num_crops) * 576 * 588);
        std::ifstream f(argv[2], std::ios::binary);
        if (!f.read(reinterpret_cast<char*>(patches.data()), patches.size() * sizeof(float))) throw Panic("patches file too short");
        auto [patch_embeddings, patch_emb_shape] = vision.forward(patches, num_crops);                        // :106
        const std::vector<float> image_features = processor.forward(patch_embeddings, patch_emb_shape, 336 / 14 / 2, w_crop, h_crop);   // :108
        std::vector<float> prefix = model.get_embeddings({1, 32010, 29871, 13});                              // :110
        const std::vector<float> suffix = model.get_embeddings({1, 29871, 13});                               // :112
        prefix.insert(prefix.end(), image_features.begin(), image_features.end());
        prefix.insert(prefix.end(), suffix.begin(), suffix.end());
        const std::uint32_t pos = model.fill_kv_cache(prefix, 0);                                                   // :119
        std::printf("image: %zu embeddings, cache filled to position %u\n", image_features.size() / model.args.dim, pos);
        std::vector<std::uint32_t> prompt;                                                                    // the user's text after the image (:148-187)
        for (int i = 7; i < argc; ++i) prompt.push_back(static_cast<std::uint32_t>(std::strtoul(argv[i], nullptr, 10)));
        for (auto t : model.generate_greedy(prompt, n_new, pos)) std::printf("%u ", t);                         // :188-222 at temperature 0
        std::printf("\n");
    } catch (const lmrs_host::Panic& e) { std::fprintf(stderr, "panic: %s\n", e.what()); return 101; }
    return 0;
}
