// chat.cpp — the reference's chat binary (src/bin/chat.rs) over the C++ mirrors, text in / text out: the same command line
// (--model, --tokenizer, --temperature, --top-p, --seed, --show-metrics), the same loop (:148-227): read a line, wrap it in the
// model family's chat template, feed the prompt token by token, sample until EOS, print the pieces.  Llama's system prompt with
// today's date (:159-169) is reproduced; --date "23 Sep 2024" pins it (the reference always uses the clock), which makes runs
// repeatable.  Images (--image) need PHI3VProcessor::process, host image code outside this library: see image_prefill.cpp.
// At temperature 0 the sampler's argmax runs on the device (forward_argmax): no logits leave HBM.
//   g++ -O2 -std=c++17 chat.cpp -I../../include -L.. -llmrs_hip -Wl,-rpath,'$ORIGIN/..' -o chat
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <iostream>

#include "text.hpp"

using namespace lmrs_host;

int main(int argc, char** argv) {
    std::string model_path, tokenizer_path = "tokenizer.bin", date;
    float temperature = 0.7f, top_p = 0.9f;
    bool have_seed = false, show_metrics = false; std::uint64_t seed = 0;
    long max_tokens = -1;                                   // (not in the reference) stop after this many sampled tokens in total: for tests
    for (int i = 1; i < argc; ++i) {
        const std::string a = argv[i];
        auto val = [&]() -> const char* { if (i + 1 >= argc) { std::fprintf(stderr, "missing value for %s\n", a.c_str()); std::exit(2); } return argv[++i]; };
        if (a == "--model") model_path = val();
        else if (a == "--tokenizer") tokenizer_path = val();
        else if (a == "--temperature") temperature = std::strtof(val(), nullptr);
        else if (a == "--top-p") top_p = std::strtof(val(), nullptr);
        else if (a == "--seed") { seed = std::strtoull(val(), nullptr, 10); have_seed = true; }
        else if (a == "--show-metrics") show_metrics = true;
        else if (a == "--date") date = val();
        else if (a == "--max-tokens") max_tokens = std::atol(val());
        else { std::fprintf(stderr, "unknown argument %s\n", a.c_str()); return 2; }
    }
    if (model_path.empty()) { std::fprintf(stderr, "usage: %s --model model.lmrs [--tokenizer tokenizer.bin] [--temperature T] [--top-p P] [--seed S] [--show-metrics]\n", argv[0]); return 2; }
    try {
        Tokenizer tokenizer(tokenizer_path);                                                      // chat.rs:60
        const int fd = open(model_path.c_str(), O_RDONLY);
        if (fd < 0) throw Panic("Error opening model file");
        struct stat st; fstat(fd, &st);
        void* m = mmap(nullptr, st.st_size, PROT_READ, MAP_PRIVATE, fd, 0);                       // :62-63
        if (m == MAP_FAILED) throw Panic("MMap failed");
        auto [model, used] = Transformer::create(static_cast<const std::uint8_t*>(m), st.st_size);   // :65
        (void)used;
        const ModelType mt = static_cast<ModelType>(model.args.model_type);
        if (!have_seed)                                                                          // :125-135
            seed = (std::uint64_t)std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::system_clock::now().time_since_epoch()).count();
        Sampler sampler(model.args.vocab_size, temperature, top_p, seed);                         // :137
        std::uint32_t pos = 0, token = 0, next = 0;
        bool user_turn = true;
        std::size_t user_idx = 0, num_prompt_tokens = 0;
        float total_tokens = 0.0f, total_duration = 0.0f;
        std::vector<std::uint32_t> prompt_tokens;
        long sampled = 0;
        for (;;) {
            if (user_turn) {
                std::printf("You: "); std::fflush(stdout);
                std::string user_prompt;
                if (!std::getline(std::cin, user_prompt)) break;                                  // (the reference would spin on EOF)
                if (mt == ModelType::LLAMA && pos == 0) {                                         // :159-169 system prompt with today's date
                    const std::uint32_t head[] = {128000, 128006, 9125, 128007, 271, 38766, 1303, 33025, 2696, 25, 6790, 220, 2366, 18, 198, 15724, 2696, 25, 220};
                    prompt_tokens.insert(prompt_tokens.end(), std::begin(head), std::end(head));
                    if (date.empty()) { char buf[32]; const std::time_t t = std::time(nullptr); std::strftime(buf, sizeof buf, "%d %b %Y", std::localtime(&t)); date = buf; }
                    const auto d = tokenizer.encode(date, false, false, false, mt);
                    prompt_tokens.insert(prompt_tokens.end(), d.begin(), d.end());
                    prompt_tokens.push_back(271); prompt_tokens.push_back(128009);
                }
                // trim() as str::trim: leading / trailing whitespace
                const auto b = user_prompt.find_first_not_of(" \t\r\n\v\f"), e = user_prompt.find_last_not_of(" \t\r\n\v\f");
                const std::string trimmed = b == std::string::npos ? std::string() : user_prompt.substr(b, e - b + 1);
                const auto enc = tokenizer.encode(trimmed, false, false, true, mt);               // :180
                prompt_tokens.insert(prompt_tokens.end(), enc.begin(), enc.end());
                num_prompt_tokens = prompt_tokens.size();
                user_turn = false; user_idx = 0;
                std::printf("Assistant:\n");
            }
            if (user_idx < num_prompt_tokens) token = prompt_tokens[user_idx++];                  // :188-193
            else token = next;
            if (token == tokenizer.eos && user_idx >= num_prompt_tokens) {                        // :195-210
                user_turn = true;
                std::printf("\n");
                prompt_tokens.clear();
                if (show_metrics) { std::printf("Speed: %.2f tok/s\n", total_tokens / (total_duration / 1000.0f)); total_duration = 0.0f; total_tokens = 0.0f; }
                continue;
            }
            const auto t0 = std::chrono::steady_clock::now();
            if (temperature == 0.0f) next = model.forward_argmax(token, pos);                     // :214-215, argmax fused on the device
            else next = sampler.forward_sample(model, token, pos);   // :115-128: scaling, maximum and exponentials on the device, the two sequential chains and sample_mult / sample_topp on the host
            pos += 1;
            if (user_idx >= num_prompt_tokens && next != tokenizer.eos && !(mt == ModelType::GEMMA && next == 107)) {   // :218-222
                std::fputs(tokenizer.decode(next).c_str(), stdout); std::fflush(stdout);
            }
            total_duration += (float)std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::steady_clock::now() - t0).count();
            total_tokens += 1.0f;
            if (user_idx >= num_prompt_tokens && max_tokens >= 0 && ++sampled >= max_tokens) { std::printf("\n"); break; }
        }
    } catch (const Panic& e) { std::fprintf(stderr, "panic: %s\n", e.what()); return 101; }
    return 0;
}
