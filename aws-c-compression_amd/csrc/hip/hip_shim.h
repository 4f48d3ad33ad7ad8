#ifndef HUFFMAN_AMD_HIP_SHIM_H
#define HUFFMAN_AMD_HIP_SHIM_H
/*
 * The thin extern "C" face of the HIP runtime that the C99 host layer uses:
 * device memory, copies, streams, events.  Streams and events travel as void *.
 * Every int return is 0 or a hipError_t (see hufs_error_string).
 */
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

int hufs_device_count(void);
int hufs_get_device(int *device);
int hufs_set_device(int device);
const char *hufs_error_string(int error);

void *hufs_malloc(size_t size); /* NULL on failure */
void hufs_free(void *ptr);
void *hufs_host_alloc(size_t size); /* page-locked host memory, NULL on failure */
void hufs_host_free(void *ptr);
int hufs_copy_h2d(void *dst, const void *src, size_t size, void *stream);
int hufs_copy_d2h(void *dst, const void *src, size_t size, void *stream);
int hufs_memset(void *dst, int byte, size_t size, void *stream);

int hufs_stream_create(void **stream);
int hufs_stream_destroy(void *stream);
int hufs_device_sync(void); /* every stream of the current device */
int hufs_stream_sync(void *stream);

void *hufs_event_create(void);
void hufs_event_destroy(void *event);
int hufs_event_record(void *event, void *stream);
void *hufs_event_create_untimed(void); /* an event that is only waited for */
int hufs_event_sync(void *event);
int hufs_event_elapsed_ms(void *start, void *stop, float *ms);

#ifdef __cplusplus
}
#endif
#endif /* HUFFMAN_AMD_HIP_SHIM_H */
