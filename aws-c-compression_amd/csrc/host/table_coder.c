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

void aws_huffman_amd_table_coder_destroy(struct aws_huffman_symbol_coder *coder) {
    if (coder) {
        free(coder->userdata);
    }
}
