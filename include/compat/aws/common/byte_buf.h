#ifndef AWS_HUFFMAN_AMD_COMPAT_BYTE_BUF_H
#define AWS_HUFFMAN_AMD_COMPAT_BYTE_BUF_H
/*
 * Stand-in for aws/common/byte_buf.h: the two buffer views that cross the
 * Huffman C ABI, with the handful of helpers the path and its tests use.
 * See common.h in this directory for when this header is (not) used.
 *
 * Layouts (x86-64 LP64) are part of the drop-in boundary:
 *   aws_byte_cursor  16 B : len@0  ptr@8
 *   aws_byte_buf     32 B : len@0  buffer@8  capacity@16  allocator@24
 */

#include <aws/common/common.h>

#include <stdlib.h>

struct aws_byte_buf {
    size_t len;      /* bytes already written */
    uint8_t *buffer; /* start of storage */
    size_t capacity; /* bytes of storage */
    struct aws_allocator *allocator; /* NULL for caller-owned arrays */
};

struct aws_byte_cursor {
    size_t len;   /* bytes left */
    uint8_t *ptr; /* next byte */
};

static inline bool aws_byte_cursor_is_valid(const struct aws_byte_cursor *cursor) {
    return cursor != NULL && (cursor->len == 0 || cursor->ptr != NULL);
}

static inline bool aws_byte_buf_is_valid(const struct aws_byte_buf *buf) {
    return buf != NULL && buf->len <= buf->capacity && (buf->capacity == 0 || buf->buffer != NULL);
}

static inline struct aws_byte_cursor aws_byte_cursor_from_array(const void *bytes, size_t len) {
    struct aws_byte_cursor c;
    c.len = len;
    c.ptr = (uint8_t *)bytes;
    return c;
}

static inline struct aws_byte_cursor aws_byte_cursor_from_buf(const struct aws_byte_buf *buf) {
    struct aws_byte_cursor c;
    c.len = buf->len;
    c.ptr = buf->buffer;
    return c;
}

static inline struct aws_byte_buf aws_byte_buf_from_empty_array(const void *bytes, size_t capacity) {
    struct aws_byte_buf b;
    b.len = 0;
    b.buffer = (uint8_t *)bytes;
    b.capacity = capacity;
    b.allocator = NULL;
    return b;
}

/* Splits off the first `len` bytes; an over-long request yields an empty cursor. */
static inline struct aws_byte_cursor aws_byte_cursor_advance(struct aws_byte_cursor *cursor, size_t len) {
    struct aws_byte_cursor head;
    if (len > cursor->len) {
        head.len = 0;
        head.ptr = NULL;
        return head;
    }
    head.len = len;
    head.ptr = cursor->ptr;
    cursor->ptr = cursor->ptr ? cursor->ptr + len : NULL;
    cursor->len -= len;
    return head;
}

static inline bool aws_byte_cursor_read_u8(struct aws_byte_cursor *cursor, uint8_t *out) {
    if (cursor->len == 0) {
        return false;
    }
    *out = *cursor->ptr;
    cursor->ptr += 1;
    cursor->len -= 1;
    return true;
}

static inline bool aws_byte_buf_write_u8(struct aws_byte_buf *buf, uint8_t value) {
    if (buf->len >= buf->capacity) {
        return false;
    }
    buf->buffer[buf->len++] = value;
    return true;
}

static inline void aws_byte_buf_reset(struct aws_byte_buf *buf, bool zero_contents) {
    if (zero_contents && buf->buffer) {
        memset(buf->buffer, 0, buf->capacity);
    }
    buf->len = 0;
}

static inline int aws_byte_buf_init(struct aws_byte_buf *buf, struct aws_allocator *allocator, size_t capacity) {
    buf->buffer = capacity ? (uint8_t *)allocator->mem_acquire(allocator, capacity) : NULL;
    if (capacity && !buf->buffer) {
        AWS_ZERO_STRUCT(*buf);
        return aws_raise_error(AWS_ERROR_OOM);
    }
    buf->len = 0;
    buf->capacity = capacity;
    buf->allocator = allocator;
    return AWS_OP_SUCCESS;
}

static inline void aws_byte_buf_clean_up(struct aws_byte_buf *buf) {
    if (buf->allocator && buf->buffer) {
        buf->allocator->mem_release(buf->allocator, buf->buffer);
    }
    AWS_ZERO_STRUCT(*buf);
}

/* Make room for `additional` more bytes past len; a no-op when they already fit. */
static inline int aws_byte_buf_reserve_relative(struct aws_byte_buf *buf, size_t additional) {
    size_t wanted = buf->len + additional;
    if (wanted < buf->len) {
        return aws_raise_error(AWS_ERROR_OOM);
    }
    if (wanted <= buf->capacity) {
        return AWS_OP_SUCCESS;
    }
    if (!buf->allocator) {
        return aws_raise_error(AWS_ERROR_INVALID_ARGUMENT);
    }
    uint8_t *grown =
        (uint8_t *)buf->allocator->mem_realloc(buf->allocator, buf->buffer, buf->capacity, wanted);
    if (!grown) {
        return aws_raise_error(AWS_ERROR_OOM);
    }
    buf->buffer = grown;
    buf->capacity = wanted;
    return AWS_OP_SUCCESS;
}

#endif /* AWS_HUFFMAN_AMD_COMPAT_BYTE_BUF_H */
