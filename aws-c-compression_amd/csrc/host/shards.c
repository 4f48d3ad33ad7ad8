/*
 * Independent items sharded over the GPUs of one node (huffman_amd.h "several GPUs"):
 * item i of a call on shard i mod G, the per-item result records gathered by the host in item order.  No collective
 * and no peer traffic: the reference's items share nothing (include/aws/compression/huffman.h:133-152 -- every call
 * has its own encoder / decoder state, cursor and buffer), so neither do the shards.
 *
 * A shard = one device with the coder's tables staged on it: an engine, its stream, ONE host thread that lives as long
 * as the shard set (it makes the device its own once and waits for work: no thread is started inside a call), and an
 * encode and a decode plan that are kept between calls -- a call with the items of the call before (the steady state of
 * a server that encodes the same batch shape again and again) uploads nothing and allocates nothing, other items reuse
 * the plans' device arrays where they are large enough.  The calling thread's current device is never touched.
 */
#include "engine.h"

#include <pthread.h>
#include <stdlib.h>
#include <string.h>

struct shard_job {
    bool decode;
    const void *items; /* the call's items, all of them */
    size_t item_count;
    const struct aws_huffman_amd_shard_io *io;
    void *results; /* the call's results, all of them */
};

struct shard {
    struct aws_huffman_amd_shards *set;
    size_t g;
    struct aws_huffman_amd_engine *engine;
    pthread_t thread;
    bool started;
    int device;
    pthread_mutex_t device_lock; /* of the first shard on a device: shards that share a device take turns (below) */
    pthread_mutex_t *turn;
    /* what the thread keeps between calls */
    struct aws_huffman_amd_encode_plan *enc_plan;
    struct aws_huffman_amd_decode_plan *dec_plan;
    void *enc_items, *dec_items; /* the items the plans were last filled with */
    size_t enc_count, dec_count;
    void *own, *res; /* this shard's share of a call's items and records */
    size_t own_bytes, res_bytes;
    /* hand-over: the caller fills `job` and raises `pending`, the thread lowers it when `rc` / `error` are set */
    struct shard_job job;
    bool pending, quit;
    int rc, error;
};

struct aws_huffman_amd_shards {
    size_t count;
    struct shard *shards;
    pthread_mutex_t lock;
    pthread_cond_t work, done;
    pthread_mutex_t call_lock; /* one call at a time per shard set */
};

static bool grow(void **buf, size_t *have, size_t want) {
    if (want > *have) {
        void *bigger = realloc(*buf, want);
        if (!bigger) {
            return false;
        }
        *buf = bigger;
        *have = want;
    }
    return true;
}

/* this shard's share of one call: its plan (kept, refilled when the items are others than last time), one launch, the records */
static void shard_work(struct shard *sh, const struct shard_job *job) {
    const size_t G = sh->set->count;
    const size_t mine = job->item_count > sh->g ? (job->item_count - sh->g + G - 1) / G : 0;
    sh->rc = AWS_OP_SUCCESS;
    sh->error = 0;
    if (mine == 0) {
        return;
    }
    const struct aws_huffman_amd_shard_io *io = &job->io[sh->g];
    const size_t item_size =
        job->decode ? sizeof(struct aws_huffman_amd_decode_item) : sizeof(struct aws_huffman_amd_encode_item);
    const size_t result_size =
        job->decode ? sizeof(struct aws_huffman_amd_decode_result) : sizeof(struct aws_huffman_amd_encode_result);
    bool ok = grow(&sh->own, &sh->own_bytes, mine * item_size) && grow(&sh->res, &sh->res_bytes, mine * result_size);
    if (!ok) {
        sh->rc = AWS_OP_ERR;
        sh->error = AWS_ERROR_OOM;
        return;
    }
    for (size_t k = 0; k < mine; ++k) {
        memcpy((char *)sh->own + k * item_size, (const char *)job->items + (sh->g + k * G) * item_size, item_size);
    }
    /* Shards that were given the same device work one after the other: the one-pass encoder sizes its grid for the whole
     * device and its workgroups wait for each other -- two such grids side by side are not resident as a whole, the
     * waits run out (milliseconds) and the launches are done over by the three-kernel road.  Correct, and slow. */
    pthread_mutex_lock(sh->turn);
    void **kept = job->decode ? &sh->dec_items : &sh->enc_items;
    size_t *kept_count = job->decode ? &sh->dec_count : &sh->enc_count;
    const bool same = *kept && *kept_count == mine && memcmp(*kept, sh->own, mine * item_size) == 0;
    if (!job->decode) {
        if (!sh->enc_plan) {
            ok = aws_huffman_amd_encode_plan_new(&sh->enc_plan, sh->engine, sh->own, mine) == AWS_OP_SUCCESS;
        } else if (!same) {
            ok = aws_huffman_amd_encode_plan_reset(sh->enc_plan, sh->own, mine) == AWS_OP_SUCCESS;
        }
        ok = ok &&
             aws_huffman_amd_encode_plan_launch(sh->enc_plan, io->device_input, io->device_output, false, NULL) == AWS_OP_SUCCESS &&
             aws_huffman_amd_encode_plan_results(sh->enc_plan, sh->res, NULL) == AWS_OP_SUCCESS;
    } else {
        if (!sh->dec_plan) {
            ok = aws_huffman_amd_decode_plan_new(&sh->dec_plan, sh->engine, sh->own, mine) == AWS_OP_SUCCESS;
        } else if (!same) {
            ok = aws_huffman_amd_decode_plan_reset(sh->dec_plan, sh->own, mine) == AWS_OP_SUCCESS;
        }
        ok = ok && aws_huffman_amd_decode_plan_launch(sh->dec_plan, io->device_input, io->device_output, NULL) == AWS_OP_SUCCESS &&
             aws_huffman_amd_decode_plan_results(sh->dec_plan, sh->res, NULL) == AWS_OP_SUCCESS;
    }
    pthread_mutex_unlock(sh->turn);
    if (!ok) {
        sh->rc = AWS_OP_ERR;
        sh->error = aws_last_error();
        *kept_count = 0; /* (whatever the plan holds now is not to be taken for these items again) */
        return;
    }
    if (!same) {
        size_t had = *kept ? *kept_count * item_size : 0;
        if (grow(kept, &had, mine * item_size)) {
            memcpy(*kept, sh->own, mine * item_size);
            *kept_count = mine;
        } else {
            *kept_count = 0;
        }
    }
    for (size_t k = 0; k < mine; ++k) {
        memcpy((char *)job->results + (sh->g + k * G) * result_size, (const char *)sh->res + k * result_size, result_size);
    }
}

static void *shard_thread(void *arg) {
    struct shard *sh = arg;
    struct aws_huffman_amd_shards *s = sh->set;
    pthread_mutex_lock(&s->lock);
    for (;;) {
        while (!sh->pending && !sh->quit) {
            pthread_cond_wait(&s->work, &s->lock);
        }
        if (sh->quit) {
            break;
        }
        const struct shard_job job = sh->job;
        pthread_mutex_unlock(&s->lock);
        shard_work(sh, &job);
        pthread_mutex_lock(&s->lock);
        sh->pending = false;
        pthread_cond_broadcast(&s->done);
    }
    pthread_mutex_unlock(&s->lock);
    return NULL;
}

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
        s->shards = calloc(device_count, sizeof(*s->shards));
    }
    if (!s || !s->shards) {
        free(s);
        return aws_raise_error(AWS_ERROR_OOM);
    }
    s->count = device_count;
    pthread_mutex_init(&s->lock, NULL);
    pthread_mutex_init(&s->call_lock, NULL);
    pthread_cond_init(&s->work, NULL);
    pthread_cond_init(&s->done, NULL);
    for (size_t g = 0; g < device_count; ++g) {
        struct shard *sh = &s->shards[g];
        sh->set = s;
        sh->g = g;
        sh->device = devices[g];
        pthread_mutex_init(&sh->device_lock, NULL);
        sh->turn = &sh->device_lock;
        for (size_t f = 0; f < g; ++f) {
            if (s->shards[f].device == devices[g]) {
                sh->turn = &s->shards[f].device_lock;
                break;
            }
        }
        /* (the engine switches to its device for its own calls and leaves this thread where it was) */
        if (aws_huffman_amd_engine_new(&sh->engine, coder, devices[g])) {
            aws_huffman_amd_shards_destroy(s);
            return AWS_OP_ERR;
        }
        sh->started = pthread_create(&sh->thread, NULL, shard_thread, sh) == 0;
        if (!sh->started) {
            aws_huffman_amd_shards_destroy(s);
            return aws_raise_error(AWS_ERROR_OOM);
        }
    }
    *out = s;
    return AWS_OP_SUCCESS;
}

void aws_huffman_amd_shards_destroy(struct aws_huffman_amd_shards *s) {
    if (!s) {
        return;
    }
    pthread_mutex_lock(&s->lock);
    for (size_t g = 0; g < s->count; ++g) {
        s->shards[g].quit = true;
    }
    pthread_cond_broadcast(&s->work);
    pthread_mutex_unlock(&s->lock);
    for (size_t g = 0; g < s->count; ++g) {
        struct shard *sh = &s->shards[g];
        if (sh->started) {
            pthread_join(sh->thread, NULL);
        }
        aws_huffman_amd_encode_plan_destroy(sh->enc_plan);
        aws_huffman_amd_decode_plan_destroy(sh->dec_plan);
        aws_huffman_amd_engine_destroy(sh->engine);
        free(sh->enc_items);
        free(sh->dec_items);
        free(sh->own);
        free(sh->res);
        if (sh->turn) {
            pthread_mutex_destroy(&sh->device_lock);
        }
    }
    pthread_cond_destroy(&s->work);
    pthread_cond_destroy(&s->done);
    pthread_mutex_destroy(&s->lock);
    pthread_mutex_destroy(&s->call_lock);
    free(s->shards);
    free(s);
}

size_t aws_huffman_amd_shards_count(const struct aws_huffman_amd_shards *s) {
    return s->count;
}

struct aws_huffman_amd_engine *aws_huffman_amd_shards_engine(struct aws_huffman_amd_shards *s, size_t g) {
    return g < s->count ? s->shards[g].engine : NULL;
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
    pthread_mutex_lock(&s->call_lock);
    pthread_mutex_lock(&s->lock);
    for (size_t g = 0; g < s->count; ++g) {
        struct shard *sh = &s->shards[g];
        sh->job.decode = decode;
        sh->job.items = items;
        sh->job.item_count = item_count;
        sh->job.io = io;
        sh->job.results = results;
        sh->pending = true;
    }
    pthread_cond_broadcast(&s->work);
    int rc = AWS_OP_SUCCESS, error = 0;
    for (size_t g = 0; g < s->count; ++g) {
        struct shard *sh = &s->shards[g];
        while (sh->pending) {
            pthread_cond_wait(&s->done, &s->lock);
        }
        if (sh->rc != AWS_OP_SUCCESS && rc == AWS_OP_SUCCESS) {
            rc = AWS_OP_ERR;
            error = sh->error;
        }
    }
    pthread_mutex_unlock(&s->lock);
    pthread_mutex_unlock(&s->call_lock);
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
