#ifndef AWS_COMPRESSION_EXPORTS_H
#define AWS_COMPRESSION_EXPORTS_H
/*
 * Symbol visibility for libaws-c-compression-amd.
 * Counterpart of reference include/aws/compression/exports.h:7-25: every public
 * entry point carries AWS_COMPRESSION_API.  This build targets Linux/ELF only
 * (ROCm), so the macro reduces to default visibility.
 */
#if defined(__GNUC__) || defined(__clang__)
#    define AWS_COMPRESSION_API __attribute__((visibility("default")))
#else
#    define AWS_COMPRESSION_API
#endif
#endif /* AWS_COMPRESSION_EXPORTS_H */
