// RawArray (.ra) I/O and IEEE half conversions -- see include/rawarray.h for the contract and
// the reference lines (src/ra.h, src/ra.cu, src/float16.cu of davidssmith/TRON) each piece follows.
#include "../../include/rawarray.h"

#include <errno.h>
#include <fcntl.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/stat.h>
#include <sysexits.h>
#include <unistd.h>

#include <vector>

namespace {

// read()/write() everything, in requests no larger than RA_MAX_BYTES (src/ra.h:59), retrying on
// short transfers.  Returns 0 or an errno-style code.
int read_all(int fd, void *buf, uint64_t count)
{
    uint8_t *p = static_cast<uint8_t *>(buf);
    while (count > 0) {
        const size_t want = count < RA_MAX_BYTES ? (size_t)count : (size_t)RA_MAX_BYTES;
        const ssize_t got = read(fd, p, want);
        if (got < 0) {
            if (errno == EINTR) continue;
            return errno;
        }
        if (got == 0) return EX_IOERR;   // premature end of file
        p += got;
        count -= (uint64_t)got;
    }
    return 0;
}

int write_all(int fd, const void *buf, uint64_t count)
{
    const uint8_t *p = static_cast<const uint8_t *>(buf);
    while (count > 0) {
        const size_t want = count < RA_MAX_BYTES ? (size_t)count : (size_t)RA_MAX_BYTES;
        const ssize_t put = write(fd, p, want);
        if (put < 0) {
            if (errno == EINTR) continue;
            return errno;
        }
        p += put;
        count -= (uint64_t)put;
    }
    return 0;
}

int read_header_fd(int fd, ra_t *a)
{
    uint64_t head[6];
    int rc = read_all(fd, head, sizeof(head));
    if (rc) {
        fprintf(stderr, "RawArray: truncated header.\n");
        return EX_IOERR;
    }
    if (head[0] != RA_MAGIC_NUMBER) {
        fprintf(stderr, "Invalid RA file.\n");                       // message of src/ra.cu:60
        return EX_DATAERR;
    }
    a->flags = head[1];
    a->eltype = head[2];
    a->elbyte = head[3];
    a->size = head[4];
    a->ndims = head[5];
    if (a->flags & ~(RA_FLAG_BIG_ENDIAN | RA_FLAG_COMPRESSED)) {     // src/ra.cu:96-100
        fprintf(stderr, "Warning: This RA file must have been written by a newer version of this\n");
        fprintf(stderr, "code. Correctness of input is not guaranteed. Update your version of the\n");
        fprintf(stderr, "RawArray package to stop this warning.\n");
    }
    if (a->ndims > 1024) {
        fprintf(stderr, "RawArray: implausible ndims %llu.\n", (unsigned long long)a->ndims);
        return EX_DATAERR;
    }
    a->dims = static_cast<uint64_t *>(malloc((a->ndims ? a->ndims : 1) * sizeof(uint64_t)));
    if (!a->dims) return ENOMEM;
    rc = read_all(fd, a->dims, a->ndims * sizeof(uint64_t));
    if (rc) {
        fprintf(stderr, "RawArray: truncated dimension list.\n");
        free(a->dims);
        a->dims = nullptr;
        return EX_IOERR;
    }
    return 0;
}

uint64_t element_count(const ra_t *r)
{
    uint64_t n = 1;
    for (uint64_t i = 0; i < r->ndims; ++i) n *= r->dims[i];
    return n;
}

double load_real(const uint8_t *p, uint64_t bytes)
{
    if (bytes == 2) {
        uint16_t h;
        memcpy(&h, p, 2);
        uint64_t b = ra_half_to_double_bits(h);
        double d;
        memcpy(&d, &b, 8);
        return d;
    }
    if (bytes == 4) {
        float f;
        memcpy(&f, p, 4);
        return f;
    }
    double d;
    memcpy(&d, p, 8);
    return d;
}

void store_real(uint8_t *p, uint64_t bytes, double v, uint64_t src_bytes, const uint8_t *src)
{
    if (bytes == 2) {
        uint16_t h;
        if (src_bytes == 4) {        // float -> half rounds once, from the float (src/float16.cu:77)
            uint32_t fb;
            memcpy(&fb, src, 4);
            h = ra_float_to_half_bits(fb);
        } else {
            uint64_t db;
            memcpy(&db, &v, 8);
            h = ra_double_to_half_bits(db);
        }
        memcpy(p, &h, 2);
    } else if (bytes == 4) {
        float f = (float)v;
        memcpy(p, &f, 4);
    } else {
        memcpy(p, &v, 8);
    }
}

}  // namespace

extern "C" int ra_read_header(ra_t *a, const char *path)
{
    memset(a, 0, sizeof(*a));
    const int fd = open(path, O_RDONLY);
    if (fd == -1) {
        fprintf(stderr, "unable to open %s for reading: %s\n", path, strerror(errno));
        return errno ? errno : EX_NOINPUT;
    }
    const int rc = read_header_fd(fd, a);
    close(fd);
    return rc;
}

extern "C" int ra_read(ra_t *a, const char *path)
{
    memset(a, 0, sizeof(*a));
    const int fd = open(path, O_RDONLY);
    if (fd == -1) {
        fprintf(stderr, "unable to open %s for reading: %s\n", path, strerror(errno));
        return errno ? errno : EX_NOINPUT;
    }
    int rc = read_header_fd(fd, a);
    if (rc) {
        close(fd);
        return rc;
    }
    {   // a regular file cannot hold more payload than its length: refuse before allocating what a corrupt header asks for
        struct stat st;
        const uint64_t head_bytes = (6 + a->ndims) * sizeof(uint64_t);
        if (fstat(fd, &st) == 0 && S_ISREG(st.st_mode) && ((uint64_t)st.st_size < head_bytes || a->size > (uint64_t)st.st_size - head_bytes)) {
            fprintf(stderr, "RawArray: file holds fewer than the %llu data bytes its header declares.\n", (unsigned long long)a->size);
            close(fd);
            free(a->dims);
            a->dims = nullptr;
            return EX_IOERR;
        }
    }
    a->data = static_cast<uint8_t *>(malloc(a->size ? a->size : 1));
    if (!a->data) {
        fprintf(stderr, "unable to allocate memory for data\n");     // src/ra.cu:117
        close(fd);
        free(a->dims);
        a->dims = nullptr;
        return ENOMEM;
    }
    rc = read_all(fd, a->data, a->size);
    close(fd);
    if (rc) {
        fprintf(stderr, "RawArray: file holds fewer than the %llu data bytes its header declares.\n", (unsigned long long)a->size);
        ra_free(a);
        return EX_IOERR;
    }
    return 0;
}

extern "C" int ra_write(ra_t *a, const char *path)
{
    const int fd = open(path, O_WRONLY | O_TRUNC | O_CREAT, 0644);   // src/ra.cu:137
    if (fd == -1) {
        fprintf(stderr, "unable to open %s for writing: %s\n", path, strerror(errno));
        return errno ? errno : EX_CANTCREAT;
    }
    const uint64_t head[6] = {RA_MAGIC_NUMBER, a->flags, a->eltype, a->elbyte, a->size, a->ndims};
    int rc = write_all(fd, head, sizeof(head));
    if (!rc) rc = write_all(fd, a->dims, a->ndims * sizeof(uint64_t));
    if (!rc) rc = write_all(fd, a->data, a->size);
    if (close(fd) != 0 && !rc) rc = errno;
    if (rc) fprintf(stderr, "RawArray: short write to %s: %s\n", path, strerror(rc));
    return rc;
}

// Streaming: the header alone (the file is created / truncated to it), then ranges of the payload at their own offsets --
// what the `tron` driver uses to read spokes while the GPU plan is built and to write images while later slices are computed.
extern "C" uint64_t ra_data_offset(const ra_t *a)
{
    return (6 + a->ndims) * sizeof(uint64_t);
}

extern "C" int ra_write_header(const ra_t *a, const char *path)
{
    const int fd = open(path, O_WRONLY | O_TRUNC | O_CREAT, 0644);
    if (fd == -1) {
        fprintf(stderr, "unable to open %s for writing: %s\n", path, strerror(errno));
        return errno ? errno : EX_CANTCREAT;
    }
    const uint64_t head[6] = {RA_MAGIC_NUMBER, a->flags, a->eltype, a->elbyte, a->size, a->ndims};
    int rc = write_all(fd, head, sizeof(head));
    if (!rc) rc = write_all(fd, a->dims, a->ndims * sizeof(uint64_t));
    if (close(fd) != 0 && !rc) rc = errno;
    if (rc) fprintf(stderr, "RawArray: short write to %s: %s\n", path, strerror(rc));
    return rc;
}

extern "C" int ra_read_range(const char *path, uint64_t data_offset, uint64_t first, uint64_t count, void *dst)
{
    const int fd = open(path, O_RDONLY);
    if (fd == -1) {
        fprintf(stderr, "unable to open %s for reading: %s\n", path, strerror(errno));
        return errno ? errno : EX_NOINPUT;
    }
    int rc = 0;
    uint8_t *q = static_cast<uint8_t *>(dst);
    uint64_t done = 0;
    while (done < count && !rc) {
        const uint64_t want = count - done < RA_MAX_BYTES ? count - done : RA_MAX_BYTES;
        const ssize_t n = pread(fd, q + done, want, (off_t)(data_offset + first + done));
        if (n < 0) { if (errno == EINTR) continue; rc = errno; }
        else if (n == 0) rc = EX_IOERR;
        else done += (uint64_t)n;
    }
    close(fd);
    if (rc) fprintf(stderr, "RawArray: cannot read %llu bytes at offset %llu of %s\n", (unsigned long long)count, (unsigned long long)first, path);
    return rc;
}

extern "C" int ra_write_range(const char *path, uint64_t data_offset, uint64_t first, uint64_t count, const void *src)
{
    const int fd = open(path, O_WRONLY);
    if (fd == -1) {
        fprintf(stderr, "unable to open %s for writing: %s\n", path, strerror(errno));
        return errno ? errno : EX_CANTCREAT;
    }
    int rc = 0;
    const uint8_t *q = static_cast<const uint8_t *>(src);
    uint64_t done = 0;
    while (done < count && !rc) {
        const uint64_t want = count - done < RA_MAX_BYTES ? count - done : RA_MAX_BYTES;
        const ssize_t n = pwrite(fd, q + done, want, (off_t)(data_offset + first + done));
        if (n < 0) { if (errno == EINTR) continue; rc = errno; }
        else done += (uint64_t)n;
    }
    if (close(fd) != 0 && !rc) rc = errno;
    if (rc) fprintf(stderr, "RawArray: short write to %s: %s\n", path, strerror(rc));
    return rc;
}

extern "C" void ra_free(ra_t *a)
{
    if (!a) return;
    free(a->dims);
    free(a->data);
    a->dims = nullptr;
    a->data = nullptr;
}

extern "C" void ra_query(const char *path)
{
    static const char *names[] = {"user", "int", "uint", "float", "complex"};
    ra_t a;
    if (ra_read_header(&a, path)) return;
    printf("---\nname: %s\n", path);
    printf("endian: %s\n", (a.flags & RA_FLAG_BIG_ENDIAN) ? "big" : "little");
    printf("compressed: %s\n", (a.flags & RA_FLAG_COMPRESSED) ? "true" : "false");
    printf("type: %s%llu\n", a.eltype < 5 ? names[a.eltype] : "unknown", (unsigned long long)a.elbyte * 8);
    printf("eltype: %llu\nelbyte: %llu\nsize: %llu\ndimension: %llu\nshape:\n", (unsigned long long)a.eltype,
           (unsigned long long)a.elbyte, (unsigned long long)a.size, (unsigned long long)a.ndims);
    for (uint64_t i = 0; i < a.ndims; ++i) printf("  - %llu\n", (unsigned long long)a.dims[i]);
    printf("...\n");
    free(a.dims);
}

extern "C" int ra_reshape(ra_t *r, const uint64_t newdims[], const uint64_t ndimsnew)
{
    uint64_t n = 1;
    for (uint64_t i = 0; i < ndimsnew; ++i) n *= newdims[i];
    if (n != element_count(r)) {
        fprintf(stderr, "ra_reshape: element count differs (%llu vs %llu)\n", (unsigned long long)n, (unsigned long long)element_count(r));
        return 1;
    }
    uint64_t *d = static_cast<uint64_t *>(malloc((ndimsnew ? ndimsnew : 1) * sizeof(uint64_t)));
    if (!d) return ENOMEM;
    memcpy(d, newdims, ndimsnew * sizeof(uint64_t));
    free(r->dims);
    r->dims = d;
    r->ndims = ndimsnew;
    return 0;
}

extern "C" int ra_squash(ra_t *r)
{
    uint64_t k = 0;
    for (uint64_t i = 0; i < r->ndims; ++i)
        if (r->dims[i] != 1) r->dims[k++] = r->dims[i];
    if (k == 0 && r->ndims > 0) r->dims[k++] = 1;
    r->ndims = k;
    return (int)k;
}

extern "C" int ra_diff(const ra_t *a, const ra_t *b)
{
    if (a->flags != b->flags || a->eltype != b->eltype || a->elbyte != b->elbyte || a->size != b->size || a->ndims != b->ndims) return 1;
    if (a->ndims && memcmp(a->dims, b->dims, a->ndims * sizeof(uint64_t))) return 2;
    if (a->size && memcmp(a->data, b->data, a->size)) return 3;
    return 0;
}

// Floating-point and complex arrays between 2-, 4- and 8-byte reals.  Other requests are ignored
// with a message (the reference declares this function and defines nothing, src/ra.h:109).
extern "C" void ra_convert(ra_t *r, const uint64_t eltype, const uint64_t elbyte)
{
    const bool is_c = r->eltype == RA_TYPE_COMPLEX;
    if ((r->eltype != RA_TYPE_FLOAT && !is_c) || eltype != r->eltype) {
        fprintf(stderr, "ra_convert: only float<->float and complex<->complex width changes are implemented\n");
        return;
    }
    const uint64_t src_real = is_c ? r->elbyte / 2 : r->elbyte;
    const uint64_t dst_real = is_c ? elbyte / 2 : elbyte;
    auto ok = [](uint64_t b) { return b == 2 || b == 4 || b == 8; };
    if (!ok(src_real) || !ok(dst_real)) {
        fprintf(stderr, "ra_convert: unsupported element width\n");
        return;
    }
    if (src_real == dst_real) return;
    const uint64_t nreal = r->size / src_real;
    uint8_t *out = static_cast<uint8_t *>(malloc(nreal * dst_real ? nreal * dst_real : 1));
    if (!out) {
        fprintf(stderr, "ra_convert: out of memory\n");
        return;
    }
    for (uint64_t i = 0; i < nreal; ++i) {
        const uint8_t *s = r->data + i * src_real;
        store_real(out + i * dst_real, dst_real, load_real(s, src_real), src_real, s);
    }
    free(r->data);
    r->data = out;
    r->elbyte = elbyte;
    r->size = nreal * dst_real;
}

// ----------------------------------------------------------------------------- binary16
// Bit-level semantics of src/float16.cu:77-324 (NumPy's halffloat.c as the reference vendors
// it): round to nearest, ties to even; NaN payloads truncated but kept NaN; overflow to
// infinity; and -- like that code -- the subnormal path shifts the significand BEFORE testing for
// a tie, so bits shifted out do not break ties (src/float16.cu:120-131).

extern "C" uint16_t ra_float_to_half_bits(uint32_t f)
{
    const uint16_t sign = (uint16_t)((f >> 16) & 0x8000u);
    const uint32_t mag = f & 0x7fffffffu;
    if (mag > 0x7f800000u) {                         // NaN
        const uint16_t payload = (uint16_t)((mag >> 13) & 0x03ffu);
        return (uint16_t)(sign | 0x7c00u | (payload ? payload : 1u));
    }
    if (mag >= 0x47800000u) return (uint16_t)(sign | 0x7c00u);   // >= 65536, and infinity
    const uint32_t e = mag >> 23;
    uint32_t sig = mag & 0x007fffffu;
    uint32_t hexp = 0;
    if (e <= 112) {                                  // below the smallest normal half
        if (e < 102) return sign;                    // below 2^-25: signed zero
        sig = (sig | 0x00800000u) >> (113 - e);
    } else {
        hexp = (e - 112) << 10;
    }
    if ((sig & 0x00003fffu) != 0x00001000u) sig += 0x00001000u;  // round half to even
    return (uint16_t)(sign + hexp + (sig >> 13));    // a carry out of the significand bumps the exponent
}

extern "C" uint16_t ra_double_to_half_bits(uint64_t d)
{
    const uint16_t sign = (uint16_t)((d >> 48) & 0x8000u);
    const uint64_t mag = d & 0x7fffffffffffffffULL;
    if (mag > 0x7ff0000000000000ULL) {
        const uint16_t payload = (uint16_t)((mag >> 42) & 0x03ffu);
        return (uint16_t)(sign | 0x7c00u | (payload ? payload : 1u));
    }
    if (mag >= 0x40f0000000000000ULL) return (uint16_t)(sign | 0x7c00u);
    const uint64_t e = mag >> 52;
    uint64_t sig = mag & 0x000fffffffffffffULL;
    uint64_t hexp = 0;
    if (e <= 1008) {
        if (e < 998) return sign;
        sig = (sig | 0x0010000000000000ULL) >> (1009 - e);
    } else {
        hexp = (e - 1008) << 10;
    }
    if ((sig & 0x000007ffffffffffULL) != 0x0000020000000000ULL) sig += 0x0000020000000000ULL;
    return (uint16_t)(sign + hexp + (sig >> 42));
}

extern "C" uint32_t ra_half_to_float_bits(uint16_t h)
{
    const uint32_t sign = ((uint32_t)h & 0x8000u) << 16;
    uint32_t e = (h >> 10) & 0x1fu;
    uint32_t m = h & 0x03ffu;
    if (e == 0x1fu) return sign | 0x7f800000u | (m << 13);
    if (e == 0) {
        if (m == 0) return sign;
        e = 1;
        while (!(m & 0x0400u)) {                     // normalise a subnormal half
            m <<= 1;
            --e;
        }
        m &= 0x03ffu;
        return sign | ((e + 112u) << 23) | (m << 13);
    }
    return sign | ((e + 112u) << 23) | (m << 13);
}

extern "C" uint64_t ra_half_to_double_bits(uint16_t h)
{
    const uint64_t sign = ((uint64_t)h & 0x8000u) << 48;
    int64_t e = (h >> 10) & 0x1f;
    uint64_t m = h & 0x03ffu;
    if (e == 0x1f) return sign | 0x7ff0000000000000ULL | (m << 42);
    if (e == 0) {
        if (m == 0) return sign;
        e = 1;
        while (!(m & 0x0400u)) {
            m <<= 1;
            --e;
        }
        m &= 0x03ffu;
    }
    return sign | ((uint64_t)(e + 1008) << 52) | (m << 42);
}
