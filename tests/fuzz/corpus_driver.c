/*
 * TEST INFRASTRUCTURE.  Runs a fuzz target (LLVMFuzzerTestOneInput of decode.c / transitive.c /
 * transitive_chunked.c) without libFuzzer: over a seeded corpus -- splitmix64 bytes, printable text, runs of one
 * symbol, the shortest and the longest codes, lengths around the sizes the kernels switch roads at -- and over
 * any files named on the command line (a crash corpus kept from a real fuzzing session replays the same way).
 *
 *   corpus_driver <inputs> <max_len> <seed> [file ...]
 *
 * Built with -fsanitize=address,undefined against the emulator build of the library (tests/emu), which runs the
 * product's own kernels on the CPU under UBSan: GPU sanitizers are not available on the pool.
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

int LLVMFuzzerTestOneInput(const uint8_t *data, size_t size);

static uint64_t s_state;
static uint64_t next_u64(void) {
    uint64_t z = (s_state += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

int main(int argc, char **argv) {
    const size_t inputs = argc > 1 ? (size_t)strtoul(argv[1], NULL, 10) : 64;
    const size_t max_len = argc > 2 ? (size_t)strtoul(argv[2], NULL, 10) : 2048;
    s_state = argc > 3 ? strtoull(argv[3], NULL, 10) : 1;
    static const size_t edges[] = {1, 2, 3, 15, 16, 17, 127, 128, 129, 511, 512, 513, 767, 768, 769};
    uint8_t *buf = malloc(max_len + 1);
    if (!buf) {
        return 2;
    }
    for (size_t k = 0; k < inputs; ++k) {
        size_t len = k < sizeof(edges) / sizeof(edges[0]) ? edges[k] : 1 + (size_t)(next_u64() % max_len);
        len = len > max_len ? max_len : len;
        const unsigned kind = (unsigned)(k % 5);
        const uint8_t one = (uint8_t)next_u64();
        for (size_t i = 0; i < len; ++i) {
            const uint8_t r = (uint8_t)next_u64();
            buf[i] = kind == 0 ? r : kind == 1 ? (uint8_t)(32 + r % 95) : kind == 2 ? one : kind == 3 ? (uint8_t)(r % 19) : (uint8_t)(r | 0x80);
        }
        LLVMFuzzerTestOneInput(buf, len);
    }
    free(buf);
    for (int a = 4; a < argc; ++a) {
        FILE *f = fopen(argv[a], "rb");
        if (!f) {
            fprintf(stderr, "cannot open %s\n", argv[a]);
            return 2;
        }
        fseek(f, 0, SEEK_END);
        const long n = ftell(f);
        fseek(f, 0, SEEK_SET);
        uint8_t *data = malloc(n > 0 ? (size_t)n : 1);
        if (!data || fread(data, 1, (size_t)n, f) != (size_t)n) {
            return 2;
        }
        fclose(f);
        LLVMFuzzerTestOneInput(data, (size_t)n);
        free(data);
    }
    printf("%zu seeded inputs and %d file(s): no finding\n", inputs, argc > 4 ? argc - 4 : 0);
    return 0;
}
