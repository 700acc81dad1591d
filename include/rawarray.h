/*
 * rawarray.h -- RawArray (.ra) file I/O, the container format of TRON's inputs and outputs.
 *
 * Same type, same function names and same on-disk format as the reference's src/ra.h /
 * src/ra.cu (davidssmith/TRON), so code written against the reference's ra_read / ra_write /
 * ra_free links against this implementation unchanged.  The five functions the reference
 * declares but never defines (src/ra.h:102,108-111: ra_query, ra_reshape, ra_convert,
 * ra_squash, ra_diff) are implemented here.
 *
 * File layout (src/ra.h:38-48, src/ra.cu:131-162), all little-endian u64:
 *   magic 0x7961727261776172 ("rawarray"), flags, eltype, elbyte, size, ndims, dims[ndims],
 *   then `size` bytes of data, first dimension fastest.
 *
 * Differences from the reference, all in error handling: functions return non-zero instead of
 * calling exit() (src/ra.cu:56-84,93-94); ra_write handles arrays larger than 2 GiB (the
 * reference's chunk loop over-reads its last chunk, src/ra.cu:153-158); ra_free releases both
 * allocations (src/ra.cu:165-174 frees `dims` only without USE_CUDA).
 */
#ifndef RAWARRAY_H
#define RAWARRAY_H

#include <stdint.h>

#if defined(__GNUC__)
#pragma GCC visibility push(default)   /* the library is built with -fvisibility=hidden: this header IS its export list */
#endif
#ifdef __cplusplus
extern "C" {
#endif

typedef struct {
    uint64_t flags;    /* RA_FLAG_* */
    uint64_t eltype;   /* ra_type */
    uint64_t elbyte;   /* bytes per element (a complex element counts both halves) */
    uint64_t size;     /* bytes of data */
    uint64_t ndims;
    uint64_t *dims;    /* malloc'ed, ndims entries */
    uint8_t *data;     /* malloc'ed, size bytes */
} ra_t;

#define RA_MAGIC_NUMBER     0x7961727261776172ULL
#define RA_FLAG_BIG_ENDIAN  (1ULL << 0)
#define RA_FLAG_COMPRESSED  (1ULL << 1)
#define RA_MAX_BYTES        (1ULL << 31)   /* largest single read()/write() request */

typedef enum {
    RA_TYPE_USER = 0,
    RA_TYPE_INT,
    RA_TYPE_UINT,
    RA_TYPE_FLOAT,
    RA_TYPE_COMPLEX
} ra_type;

/* src/ra.cu:87-128.  Returns 0, or an errno-style code with a message on stderr. */
int ra_read(ra_t *a, const char *path);
/* src/ra.cu:131-162 */
int ra_write(ra_t *a, const char *path);
/* src/ra.cu:165-174 */
void ra_free(ra_t *a);

/* Declared at src/ra.h:102,108-111, undefined in the reference. */
void ra_query(const char *path);                                            /* prints the header */
int ra_reshape(ra_t *r, const uint64_t newdims[], const uint64_t ndimsnew); /* same element count */
void ra_convert(ra_t *r, const uint64_t eltype, const uint64_t elbyte);     /* float<->half<->double, complex likewise */
int ra_squash(ra_t *r);                                                     /* drops singleton dims; returns new ndims */
int ra_diff(const ra_t *a, const ra_t *b);                                  /* 0 if header and bytes agree */

/* Header only (no data read); dims is malloc'ed. */
int ra_read_header(ra_t *a, const char *path);

/* Streaming (not in the reference): the header alone -- the file is created or truncated to it -- and byte ranges of the
   payload, counted from its first byte; data_offset = ra_data_offset().  Each call opens the file itself (thread-safe). */
uint64_t ra_data_offset(const ra_t *a);
int ra_write_header(const ra_t *a, const char *path);
int ra_read_range(const char *path, uint64_t data_offset, uint64_t first, uint64_t count, void *dst);
int ra_write_range(const char *path, uint64_t data_offset, uint64_t first, uint64_t count, const void *src);

/* IEEE binary16 conversions with round-to-nearest-even, the semantics of src/float16.cu:42-324. */
uint16_t ra_float_to_half_bits(uint32_t f);
uint32_t ra_half_to_float_bits(uint16_t h);
uint16_t ra_double_to_half_bits(uint64_t d);
uint64_t ra_half_to_double_bits(uint16_t h);

#ifdef __cplusplus
}
#endif
#if defined(__GNUC__)
#pragma GCC visibility pop
#endif
#endif /* RAWARRAY_H */
