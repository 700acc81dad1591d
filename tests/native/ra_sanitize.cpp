// Host-only hardening check, built with -fsanitize=address,undefined by tests/test_host_sanitize.py:
// the .ra reader against truncated / hostile headers, the writers and converters on odd shapes, and the
// table builders of tron_hostmath.cpp.  Exits 0 when nothing trips a sanitizer or an assertion.
#include <assert.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

#include <string>
#include <vector>

#include "../../include/rawarray.h"
#include "../../include/tron_hip.h"

// tron_hostmath.cpp reports errors through tron::fail (defined next to the HIP code in tron_plan.cpp)
namespace tron { int fail(int code, const char *, ...) { return code; } }

static void write_bytes(const char *path, const std::vector<uint8_t> &b)
{
    FILE *f = fopen(path, "wb");
    assert(f);
    if (!b.empty()) assert(fwrite(b.data(), 1, b.size(), f) == b.size());
    fclose(f);
}

int main(int argc, char **argv)
{
    const std::string dir = argc > 1 ? argv[1] : "/tmp";
    const std::string good = dir + "/san_good.ra", bad = dir + "/san_bad.ra";

    // a valid complex64 array, 5-D
    ra_t a;
    memset(&a, 0, sizeof(a));
    uint64_t dims[5] = {2, 1, 8, 5, 1};
    a.eltype = RA_TYPE_COMPLEX; a.elbyte = 8; a.ndims = 5; a.size = 2 * 8 * 5 * 8;
    a.dims = (uint64_t *)malloc(sizeof(dims)); memcpy(a.dims, dims, sizeof(dims));
    a.data = (uint8_t *)malloc(a.size);
    for (uint64_t i = 0; i < a.size / 4; ++i) ((float *)a.data)[i] = (float)i * 0.25f - 17.f;
    assert(ra_write(&a, good.c_str()) == 0);
    ra_t b;
    assert(ra_read(&b, good.c_str()) == 0);
    assert(ra_diff(&a, &b) == 0);

    // conversions and reshapes round trip
    ra_convert(&b, RA_TYPE_COMPLEX, 4);            // complex64 -> complex-half
    assert(b.elbyte == 4 && b.size == a.size / 2);
    ra_convert(&b, RA_TYPE_COMPLEX, 16);           // -> complex128
    assert(b.elbyte == 16 && b.size == a.size * 2);
    ra_convert(&b, RA_TYPE_COMPLEX, 8);
    assert(b.size == a.size);
    uint64_t nd[2] = {16, 5};
    assert(ra_reshape(&b, nd, 2) == 0 && b.ndims == 2);
    uint64_t wrong[2] = {16, 6};
    assert(ra_reshape(&b, wrong, 2) != 0);
    assert(ra_squash(&b) == 2);
    ra_free(&b);

    // every truncation of the valid file, and a byte flip at every header offset
    FILE *f = fopen(good.c_str(), "rb");
    std::vector<uint8_t> bytes(48 + 40 + a.size);
    assert(fread(bytes.data(), 1, bytes.size(), f) == bytes.size());
    fclose(f);
    for (size_t cut = 0; cut < bytes.size(); cut += (cut < 100 ? 1 : 97)) {
        write_bytes(bad.c_str(), std::vector<uint8_t>(bytes.begin(), bytes.begin() + cut));
        ra_t c;
        if (ra_read(&c, bad.c_str()) == 0) ra_free(&c);
        ra_t h;
        if (ra_read_header(&h, bad.c_str()) == 0) free(h.dims);
    }
    for (size_t off = 0; off < 88; ++off)
        for (int bit = 0; bit < 8; bit += 3) {
            std::vector<uint8_t> m = bytes;
            m[off] ^= (uint8_t)(1u << bit);
            write_bytes(bad.c_str(), m);
            ra_t c;
            if (ra_read(&c, bad.c_str()) == 0) ra_free(&c);
        }
    // hostile header fields: huge ndims, huge size, zero dims
    {
        std::vector<uint8_t> m = bytes;
        uint64_t v = ~0ull; memcpy(&m[40], &v, 8);              // ndims
        write_bytes(bad.c_str(), m);
        ra_t c; assert(ra_read(&c, bad.c_str()) != 0);
        m = bytes; v = 1ull << 60; memcpy(&m[32], &v, 8);       // size
        write_bytes(bad.c_str(), m);
        assert(ra_read(&c, bad.c_str()) != 0);
    }

    // half conversions: exhaustive half -> float -> half, and the rounding corner cases
    for (uint32_t h = 0; h < 65536; ++h) {
        const uint32_t fb = ra_half_to_float_bits((uint16_t)h);
        const uint16_t back = ra_float_to_half_bits(fb);
        const bool nan = (h & 0x7c00) == 0x7c00 && (h & 0x3ff);
        assert(nan ? ((back & 0x7c00) == 0x7c00 && (back & 0x3ff)) : back == h);
        const uint64_t db = ra_half_to_double_bits((uint16_t)h);
        const uint16_t back2 = ra_double_to_half_bits(db);
        assert(nan ? ((back2 & 0x7c00) == 0x7c00 && (back2 & 0x3ff)) : back2 == h);
    }

    // dimension logic and host tables on awkward sizes (no device involved)
    tron_config cfg;
    for (int adj = 0; adj < 2; ++adj)
        for (uint64_t n : {1ull, 2ull, 3ull, 7ull, 64ull, 513ull})
            for (float os : {1.0f, 1.25f, 2.0f, 3.0f}) {
                tron_config_default(&cfg);
                cfg.adjoint = adj; cfg.golden_angle = 1; cfg.gridos = os; cfg.data_undersamp = 0.37f; cfg.prof_slide = 3;
                uint64_t d5[5] = {2, 1, n, adj ? 50 : n, 1};
                tron_dims td;
                if (tron_derive_dims(&cfg, d5, &td) != TRON_OK) continue;
                if (td.nxos > 0 && td.nxos <= 1024) {
                    std::vector<uint32_t> band((size_t)td.nxos * td.nxos);
                    tron_host_band_table(td.nxos, cfg.kernwidth, band.data());
                }
            }
    ra_free(&a);
    unlink(good.c_str());
    unlink(bad.c_str());
    puts("ra_sanitize: ok");
    return 0;
}
