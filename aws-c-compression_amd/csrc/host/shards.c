/*
 * Independent items sharded over the GPUs of one node (huffman_amd.h "several GPUs"):
 * one engine + stream per device, one host thread per shard inside a call, item i on shard i mod G, the per-item
 * result records gathered by the host in item order.  No collective and no peer traffic: the reference's items
 * share nothing (include/aws/compression/huffman.h:133-152 -- every call has its own encoder / decoder state,
 * cursor and buffer), so neither do the shards.
 */
#include "engine.h"

#include <pthread.h>
#include <stdlib.h>
#include <string.h>

struct aws_huffman_amd_shards {
    size_t count;
    struct aws_huffman_amd_engine **engines;
};

int aws_huffman_amd_shards_new(
    struct aws_huffman_amd_shards **out,
    struct aws_huffman_symbol_coder *coder,
    const int *devices,
    size_t device_count) {

    *out = NULL;
    if (!coder || !devices || device_count == 0) {
        return aws_raise_error(AWS_ERROR_INVALID_ARGUMENT);
    }
    struct aws_huffman_amd_shards *s = calloc(1, sizeof(*s));
    if (s) {
        s->engines = calloc(device_count, sizeof(*s->engines));
    }
    if (!s || !s->engines) {
        free(s);
        return aws_raise_error(AWS_ERROR_OOM);
    }
    s->count = device_count;
    for (size_t g = 0; g < device_count; ++g) {
        if (aws_huffman_amd_engine_new(&s->engines[g], coder, devices[g])) {
            aws_huffman_amd_shards_destroy(s);
            return AWS_OP_ERR;
        }
    }
    *out = s;
    return AWS_OP_SUCCESS;
}

void aws_huffman_amd_shards_destroy(struct aws_huffman_amd_shards *s) {
    if (!s) {
        return;
    }
    for (size_t g = 0; g < s->count; ++g) {
        aws_huffman_amd_engine_destroy(s->engines[g]);
    }
    free(s->engines);
    free(s);
}

size_t aws_huffman_amd_shards_count(const struct aws_huffman_amd_shards *s) {
    return s->count;
}

struct aws_huffman_amd_engine *aws_huffman_amd_shards_engine(struct aws_huffman_amd_shards *s, size_t g) {
    return g < s->count ? s->engines[g] : NULL;
}

/* what one shard's thread does in a call */
struct shard_job {
    struct aws_huffman_amd_shards *shards;
    size_t g;
    bool decode;
    const void *items; /* the call's items, all of them */
    size_t item_count;
    const struct aws_huffman_amd_shard_io *io;
    void *results;     /* the call's results, all of them */
    int rc;
    int error;
};

static void *shard_run(void *arg) {
    struct shard_job *job = arg;
    struct aws_huffman_amd_engine *eng = job->shards->engines[job->g];
    const size_t G = job->shards->count;
    const size_t mine = job->item_count > job->g ? (job->item_count - job->g + G - 1) / G : 0;
    job->rc = AWS_OP_SUCCESS;
    job->error = 0;
    if (mine == 0) {
        return NULL;
    }
    const struct aws_huffman_amd_shard_io *io = &job->io[job->g];
    if (!job->decode) {
        const struct aws_huffman_amd_encode_item *all = job->items;
        struct aws_huffman_amd_encode_result *out = job->results;
        struct aws_huffman_amd_encode_item *own = malloc(mine * sizeof(*own));
        struct aws_huffman_amd_encode_result *res = malloc(mine * sizeof(*res));
        struct aws_huffman_amd_encode_plan *plan = NULL;
        bool ok = own && res;
        for (size_t k = 0; ok && k < mine; ++k) {
            own[k] = all[job->g + k * G];
        }
        ok = ok && aws_huffman_amd_encode_plan_new(&plan, eng, own, mine) == AWS_OP_SUCCESS &&
             aws_huffman_amd_encode_plan_launch(plan, io->device_input, io->device_output, false, NULL) == AWS_OP_SUCCESS &&
             aws_huffman_amd_encode_plan_results(plan, res, NULL) == AWS_OP_SUCCESS;
        for (size_t k = 0; ok && k < mine; ++k) {
            out[job->g + k * G] = res[k];
        }
        if (!ok) {
            job->rc = AWS_OP_ERR;
            job->error = own && res ? aws_last_error() : AWS_ERROR_OOM;
        }
        aws_huffman_amd_encode_plan_destroy(plan);
        free(own);
        free(res);
    } else {
        const struct aws_huffman_amd_decode_item *all = job->items;
        struct aws_huffman_amd_decode_result *out = job->results;
        struct aws_huffman_amd_decode_item *own = malloc(mine * sizeof(*own));
        struct aws_huffman_amd_decode_result *res = malloc(mine * sizeof(*res));
        struct aws_huffman_amd_decode_plan *plan = NULL;
        bool ok = own && res;
        for (size_t k = 0; ok && k < mine; ++k) {
            own[k] = all[job->g + k * G];
        }
        ok = ok && aws_huffman_amd_decode_plan_new(&plan, eng, own, mine) == AWS_OP_SUCCESS &&
             aws_huffman_amd_decode_plan_launch(plan, io->device_input, io->device_output, NULL) == AWS_OP_SUCCESS &&
             aws_huffman_amd_decode_plan_results(plan, res, NULL) == AWS_OP_SUCCESS;
        for (size_t k = 0; ok && k < mine; ++k) {
            out[job->g + k * G] = res[k];
        }
        if (!ok) {
            job->rc = AWS_OP_ERR;
            job->error = own && res ? aws_last_error() : AWS_ERROR_OOM;
        }
        aws_huffman_amd_decode_plan_destroy(plan);
        free(own);
        free(res);
    }
    return NULL;
}

static int shards_call(
    struct aws_huffman_amd_shards *s,
    bool decode,
    const void *items,
    size_t item_count,
    const struct aws_huffman_amd_shard_io *io,
    void *results) {

    if (!s || (item_count && (!items || !io || !results))) {
        return aws_raise_error(AWS_ERROR_INVALID_ARGUMENT);
    }
    struct shard_job *jobs = calloc(s->count, sizeof(*jobs));
    pthread_t *threads = calloc(s->count, sizeof(*threads));
    bool *started = calloc(s->count, sizeof(*started));
    if (!jobs || !threads || !started) {
        free(jobs);
        free(threads);
        free(started);
        return aws_raise_error(AWS_ERROR_OOM);
    }
    for (size_t g = 0; g < s->count; ++g) {
        jobs[g].shards = s;
        jobs[g].g = g;
        jobs[g].decode = decode;
        jobs[g].items = items;
        jobs[g].item_count = item_count;
        jobs[g].io = io;
        jobs[g].results = results;
        /* the last shard's work on the calling thread: one thread fewer to start, and a lone shard none at all */
        if (g + 1 < s->count) {
            started[g] = pthread_create(&threads[g], NULL, shard_run, &jobs[g]) == 0;
            if (!started[g]) {
                shard_run(&jobs[g]);
            }
        } else {
            shard_run(&jobs[g]);
        }
    }
    int rc = AWS_OP_SUCCESS, error = 0;
    for (size_t g = 0; g < s->count; ++g) {
        if (started[g]) {
            pthread_join(threads[g], NULL);
        }
        if (jobs[g].rc != AWS_OP_SUCCESS && rc == AWS_OP_SUCCESS) {
            rc = AWS_OP_ERR;
            error = jobs[g].error;
        }
    }
    free(jobs);
    free(threads);
    free(started);
    return rc == AWS_OP_SUCCESS ? AWS_OP_SUCCESS : aws_raise_error(error ? error : AWS_ERROR_UNKNOWN);
}

int aws_huffman_amd_shards_encode(
    struct aws_huffman_amd_shards *s,
    const struct aws_huffman_amd_encode_item *items,
    size_t item_count,
    const struct aws_huffman_amd_shard_io *io,
    struct aws_huffman_amd_encode_result *results) {
    return shards_call(s, false, items, item_count, io, results);
}

int aws_huffman_amd_shards_decode(
    struct aws_huffman_amd_shards *s,
    const struct aws_huffman_amd_decode_item *items,
    size_t item_count,
    const struct aws_huffman_amd_shard_io *io,
    struct aws_huffman_amd_decode_result *results) {
    return shards_call(s, true, items, item_count, io, results);
}
