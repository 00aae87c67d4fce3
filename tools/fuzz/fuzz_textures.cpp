// Sanitizer run of the texture decoders on the host: every fixture of tests/golden/textures, damaged at random (byte flips, cuts,
// insertions, header words), goes through crtimg::load; the decoders must return an image or an error -- AddressSanitizer and
// UBSan watch the rest.
//   g++ -std=c++17 -O1 -g -fsanitize=address,undefined -fno-sanitize-recover=all -Icudaraytracing_amd/csrc tools/fuzz/fuzz_textures.cpp -o /tmp/fuzz_textures
//   /tmp/fuzz_textures <iterations per file> tests/golden/textures/*
#include "crt_image.h"

#include <unistd.h>

#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

static uint64_t rng_state = 0x9e3779b97f4a7c15ull;
static uint32_t rnd() { rng_state = rng_state * 6364136223846793005ull + 1442695040888963407ull; return (uint32_t)(rng_state >> 33); }

int main(int argc, char** argv)
{
    if (argc < 3) { std::fprintf(stderr, "usage: fuzz_textures <iterations> <files...>\n"); return 2; }
    const int iters = std::atoi(argv[1]);
    const std::string tmp = std::string("/dev/shm/crt_fuzz_") + std::to_string((long)getpid()) + ".bin";
    long decoded = 0, rejected = 0;
    for (int a = 2; a < argc; a++) {
        std::vector<uint8_t> orig;
        if (FILE* f = std::fopen(argv[a], "rb")) { uint8_t buf[4096]; size_t n; while ((n = std::fread(buf, 1, sizeof(buf), f)) > 0) orig.insert(orig.end(), buf, buf + n); std::fclose(f); }
        if (orig.size() < 8) continue;
        for (int it = 0; it < iters; it++) {
            std::vector<uint8_t> b = orig;
            switch (it & 3) {
            case 0: for (uint32_t k = 1 + rnd() % 4; k > 0; k--) b[rnd() % b.size()] = (uint8_t)rnd(); break;
            case 1: b.resize(1 + rnd() % (b.size() - 1)); break;
            case 2: { const size_t at = 2 + rnd() % (b.size() - 2); std::vector<uint8_t> ins(1 + rnd() % 16); for (auto& v : ins) v = (uint8_t)rnd(); b.insert(b.begin() + at, ins.begin(), ins.end()); break; }
            default: { const size_t lim = b.size() - 4 < 400 ? b.size() - 4 : 400; const size_t at = 2 + rnd() % (lim > 2 ? lim - 2 : 1); b[at] = (uint8_t)rnd(); b[at + 1] = (uint8_t)rnd(); break; }
            }
            FILE* f = std::fopen(tmp.c_str(), "wb");
            if (!f) return 3;
            std::fwrite(b.data(), 1, b.size(), f);
            std::fclose(f);
            crtimg::Image img;
            const std::string e = crtimg::load(tmp, img);
            if (e.empty()) {
                if (img.px.size() != (size_t)img.width * img.height * img.comp || img.comp < 1 || img.comp > 4) { std::fprintf(stderr, "inconsistent image from %s (iteration %d)\n", argv[a], it); return 1; }
                decoded++;
            } else rejected++;
        }
    }
    std::remove(tmp.c_str());
    std::printf("{\"decoded\": %ld, \"rejected\": %ld}\n", decoded, rejected);
    return 0;
}
