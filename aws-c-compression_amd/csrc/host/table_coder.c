/*
 * A symbol coder built at run time from 256 (pattern, num_bits) rows.
 *
 * Runtime counterpart of the reference's offline generator
 * (source/huffman_generator/generator.c): where that tool emits a C file with a
 * table-lookup encoder and a goto-tree decoder, this builds the same two
 * functions as data.  For a prefix-free code the goto tree (first leaf reached
 * wins, a missing child returns 0; generator.c:154-214) accepts exactly the
 * windows that start with one of the codes, so the decoder here keeps the codes
 * sorted by their left-aligned value and finds the only code that can match by
 * binary search instead of walking bit by bit.
 */
#include <aws/compression/huffman_amd.h>

#include <stdlib.h>
#include <string.h>

struct sorted_code {
    uint32_t left_aligned; /* code << (32 - num_bits) */
    uint8_t num_bits;
    uint8_t symbol;
};

struct table_coder {
    struct aws_huffman_symbol_coder coder;
    struct aws_huffman_code rows[256];
    struct sorted_code sorted[256];
    int n_sorted;
};

static struct aws_huffman_code table_encode(uint8_t symbol, void *userdata) {
    return ((const struct table_coder *)userdata)->rows[symbol];
}

static uint8_t table_decode(uint32_t bits, uint8_t *symbol, void *userdata) {
    const struct table_coder *tc = (const struct table_coder *)userdata;
    /* last entry whose left-aligned code is <= bits */
    int lo = 0, hi = tc->n_sorted;
    while (lo < hi) {
        const int mid = (lo + hi) / 2;
        if (tc->sorted[mid].left_aligned <= bits) {
            lo = mid + 1;
        } else {
            hi = mid;
        }
    }
    if (lo == 0) {
        return 0;
    }
    const struct sorted_code *c = &tc->sorted[lo - 1];
    if (((bits ^ c->left_aligned) >> (32 - c->num_bits)) != 0) {
        return 0;
    }
    *symbol = c->symbol;
    return c->num_bits;
}

static int by_left_aligned(const void *a, const void *b) {
    const struct sorted_code *x = a, *y = b;
    if (x->left_aligned != y->left_aligned) {
        return x->left_aligned < y->left_aligned ? -1 : 1;
    }
    return (int)x->num_bits - (int)y->num_bits;
}

struct aws_huffman_symbol_coder *aws_huffman_amd_table_coder_new(
    const uint32_t patterns[256],
    const uint8_t num_bits[256]) {

    struct table_coder *tc = calloc(1, sizeof(*tc));
    if (!tc) {
        aws_raise_error(AWS_ERROR_OOM);
        return NULL;
    }
    for (int s = 0; s < 256; ++s) {
        const uint8_t n = num_bits[s];
        if (n > 32 || (n > 0 && n < 32 && (patterns[s] >> n) != 0)) {
            free(tc);
            aws_raise_error(AWS_ERROR_INVALID_ARGUMENT);
            return NULL;
        }
        tc->rows[s].pattern = patterns[s];
        tc->rows[s].num_bits = n;
        if (n) {
            struct sorted_code *c = &tc->sorted[tc->n_sorted++];
            c->left_aligned = patterns[s] << (32 - n);
            c->num_bits = n;
            c->symbol = (uint8_t)s;
        }
    }
    qsort(tc->sorted, (size_t)tc->n_sorted, sizeof(tc->sorted[0]), by_left_aligned);
    /* prefix-free <=> the sorted code intervals do not overlap */
    for (int i = 0; i + 1 < tc->n_sorted; ++i) {
        const struct sorted_code *a = &tc->sorted[i], *b = &tc->sorted[i + 1];
        const uint64_t a_end = (uint64_t)a->left_aligned + (1ull << (32 - a->num_bits));
        if (a_end > b->left_aligned) {
            free(tc);
            aws_raise_error(AWS_ERROR_INVALID_ARGUMENT);
            return NULL;
        }
    }
    tc->coder.encode = table_encode;
    tc->coder.decode = table_decode;
    tc->coder.userdata = tc;
    return &tc->coder;
}

/* ------------------------------------------------------------------ a coder from the text of a table .def file */

/* skips blanks, comments and preprocessor lines; returns the position of the next token or `end` */
static const char *def_skip(const char *at, const char *end, bool *line_start) {
    while (at < end) {
        if (*at == '\n') {
            *line_start = true;
            ++at;
        } else if (*at == ' ' || *at == '\t' || *at == '\r') {
            ++at;
        } else if (*line_start && *at == '#') {
            while (at < end && *at != '\n') {
                ++at;
            }
        } else if (at + 1 < end && at[0] == '/' && at[1] == '*') {
            at += 2;
            while (at + 1 < end && !(at[0] == '*' && at[1] == '/')) {
                ++at;
            }
            at = at + 1 < end ? at + 2 : end;
        } else if (at + 1 < end && at[0] == '/' && at[1] == '/') {
            while (at < end && *at != '\n') {
                ++at;
            }
        } else {
            break;
        }
    }
    return at;
}

/* one unsigned number, decimal or 0x-hex (`hex`: hex even without the prefix, as the reference reads patterns) */
static bool def_number(const char **at, const char *end, bool hex, uint64_t *out) {
    const char *p = *at;
    uint64_t v = 0;
    int digits = 0;
    if (p + 1 < end && p[0] == '0' && (p[1] == 'x' || p[1] == 'X')) {
        hex = true;
        p += 2;
    }
    for (; p < end; ++p, ++digits) {
        int d;
        if (*p >= '0' && *p <= '9') {
            d = *p - '0';
        } else if (hex && *p >= 'a' && *p <= 'f') {
            d = *p - 'a' + 10;
        } else if (hex && *p >= 'A' && *p <= 'F') {
            d = *p - 'A' + 10;
        } else {
            break;
        }
        v = v * (hex ? 16u : 10u) + (uint64_t)d;
        if (v > 0xFFFFFFFFull) {
            return false;
        }
    }
    *at = p;
    *out = v;
    return digits > 0;
}

static bool def_expect(const char **at, const char *end, char c, bool *line_start) {
    *at = def_skip(*at, end, line_start);
    if (*at < end && **at == c) {
        ++*at;
        *line_start = false;
        return true;
    }
    return false;
}

struct aws_huffman_symbol_coder *aws_huffman_amd_table_coder_from_def(const char *text, size_t length) {
    static const char keyword[] = "HUFFMAN_CODE";
    const size_t keyword_len = sizeof(keyword) - 1;
    uint32_t patterns[256];
    uint8_t num_bits[256];
    bool seen[256];
    memset(patterns, 0, sizeof(patterns));
    memset(num_bits, 0, sizeof(num_bits));
    memset(seen, 0, sizeof(seen));

    const char *at = text, *end = text + length;
    bool line_start = true;
    size_t rows = 0;
    for (;;) {
        at = def_skip(at, end, &line_start);
        if (at >= end) {
            break;
        }
        if ((size_t)(end - at) < keyword_len || memcmp(at, keyword, keyword_len) != 0) {
            ++at; /* not a row: the #ifndef guard's body, stray text */
            line_start = false;
            continue;
        }
        at += keyword_len;
        line_start = false;
        uint64_t symbol = 0, pattern = 0, bits = 0;
        bool ok = def_expect(&at, end, '(', &line_start);
        at = def_skip(at, end, &line_start);
        ok = ok && def_number(&at, end, false, &symbol) && def_expect(&at, end, ',', &line_start);
        /* the bit string is for the reader: the pattern and the length are what counts (reference generator.c:84-85) */
        ok = ok && def_expect(&at, end, '"', &line_start);
        while (ok && at < end && *at != '"') {
            ++at;
        }
        ok = ok && def_expect(&at, end, '"', &line_start) && def_expect(&at, end, ',', &line_start);
        at = def_skip(at, end, &line_start);
        ok = ok && def_number(&at, end, true, &pattern) && def_expect(&at, end, ',', &line_start);
        at = def_skip(at, end, &line_start);
        ok = ok && def_number(&at, end, false, &bits) && def_expect(&at, end, ')', &line_start);
        if (!ok || bits > 32) {
            aws_raise_error(AWS_ERROR_INVALID_ARGUMENT);
            return NULL;
        }
        if (symbol > 255) {
            continue; /* HPACK's EOS (256) and the like: not a byte symbol, no row in the 256-entry coder */
        }
        if (seen[symbol]) {
            aws_raise_error(AWS_ERROR_INVALID_ARGUMENT); /* "Symbol already found!" (generator.c:78) */
            return NULL;
        }
        seen[symbol] = true;
        patterns[symbol] = (uint32_t)pattern;
        num_bits[symbol] = (uint8_t)bits;
        ++rows;
    }
    if (rows == 0) {
        aws_raise_error(AWS_ERROR_INVALID_ARGUMENT);
        return NULL;
    }
    return aws_huffman_amd_table_coder_new(patterns, num_bits);
}

/* (huffman.c: retires the engines tabulated from this coder, so that neither their device memory nor -- should the
 * allocator hand the address out again -- their tables outlive it) */
void aws_huffman_amd_forget_coder(struct aws_huffman_symbol_coder *coder);

void aws_huffman_amd_table_coder_destroy(struct aws_huffman_symbol_coder *coder) {
    if (coder) {
        aws_huffman_amd_forget_coder(coder);
        free(coder->userdata);
    }
}
