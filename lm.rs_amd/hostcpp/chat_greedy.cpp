// chat_greedy.cpp — the reference's generation loop (src/bin/chat.rs:148-227) on token ids, greedy, over
// the C++ mirror.  Tokenizer / stdin / sampler variants are out of scope, so the prompt is a
// list of token ids.   usage: chat_greedy model.lmrs N_NEW id id id ...
//   g++ -O2 -std=c++17 chat_greedy.cpp -L.. -llmrs_hip -Wl,-rpath,'$ORIGIN/..' -o chat_greedy
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <cstdio>
#include <cstdlib>

#include "transformer.hpp"

int main(int argc, char** argv) {
    if (argc < 4) { std::fprintf(stderr, "usage: %s model.lmrs n_new token_id...\n", argv[0]); return 2; }
    const int fd = open(argv[1], O_RDONLY);
    if (fd < 0) { std::perror("open"); return 1; }
    struct stat st; fstat(fd, &st);
    void* m = mmap(nullptr, st.st_size, PROT_READ, MAP_PRIVATE, fd, 0);       // chat.rs:62-64
    if (m == MAP_FAILED) { std::perror("mmap"); return 1; }
    try {
        auto [model, used] = lmrs_host::Transformer::create(static_cast<const std::uint8_t*>(m), st.st_size);   // chat.rs:65
        std::vector<std::uint32_t> prompt;
        for (int i = 3; i < argc; ++i) prompt.push_back(static_cast<std::uint32_t>(std::strtoul(argv[i], nullptr, 10)));
        double sec = 0;
        auto out = model.generate_greedy(prompt, static_cast<std::uint32_t>(std::atoi(argv[2])), 0, &sec);
        for (auto t : out) std::printf("%u ", t);
        std::printf("\nSpeed: %.2f tok/s\n", (prompt.size() + out.size() - 1) / sec);   // chat.rs:224-226 (without its ms truncation)
        (void)used;
    } catch (const lmrs_host::Panic& e) { std::fprintf(stderr, "panic: %s\n", e.what()); return 101; }
    return 0;
}
