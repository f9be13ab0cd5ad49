// tests/jpeg_assemble_fuzz.cpp — TEST INFRASTRUCTURE: the host half of the device JPEG encoder (denseflow_amd/csrc/jpeg_host.cpp:
// file header + 0xFF byte stuffing + EOI) on random bit strings under AddressSanitizer / UBSan: every buffer is exactly as large
// as the file, one byte less must be refused, no bare 0xFF may appear in the segment.  Built and run by
// tests/test_jpeg_host.py::test_assembly_fuzz_under_sanitizers; needs no GPU (jpeg_host.cpp is plain C++).
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include "../denseflow_amd/csrc/jpeg_kernels.h"
int main() {
    std::mt19937 rng(7);
    unsigned char q[64];
    JpegTables t;
    long checked = 0;
    for (int quality : {1, 50, 95, 100}) {
        jpeg_build_tables(quality, t, q);
        for (int it = 0; it < 3000; ++it) {
            const int w = 1 + rng() % 70, h = 1 + rng() % 50;
            std::vector<unsigned char> hdr = jpeg_file_header(w, h, q);
            const unsigned long long bits = rng() % 5000;
            const size_t nbytes = (size_t)((bits + 7) / 8);
            std::vector<unsigned char> src(nbytes ? nbytes : 1);
            const int mode = rng() % 3;
            for (auto &b : src) b = mode == 0 ? 0xFF : (mode == 1 ? (unsigned char)rng() : (unsigned char)((rng() % 4) ? 0xFF : rng()));
            // the bits beyond `bits` in the last byte are zero in the device stream
            if (nbytes && (bits & 7)) src[nbytes - 1] &= (unsigned char)(0xFF << (8 - (bits & 7)));
            // exact-size buffer (heap: ASan guards both ends)
            std::vector<unsigned char> big(hdr.size() + 2 * nbytes + 16);
            const size_t n = jpeg_assemble(hdr, src.data(), bits, big.data(), big.size());
            if (n == 0) { printf("unexpected failure\n"); return 1; }
            std::vector<unsigned char> exact(n);
            if (jpeg_assemble(hdr, src.data(), bits, exact.data(), n) != n || memcmp(exact.data(), big.data(), n)) { printf("exact differs\n"); return 1; }
            if (n > 1) { std::vector<unsigned char> small(n - 1); if (jpeg_assemble(hdr, src.data(), bits, small.data(), n - 1) != 0) { printf("overflow accepted\n"); return 1; } }
            // structure: header, no bare 0xFF in the segment, EOI
            if (exact[n - 2] != 0xFF || exact[n - 1] != 0xD9) { printf("no EOI\n"); return 1; }
            for (size_t i = hdr.size(); i + 2 < n; ++i) if (exact[i] == 0xFF && exact[i + 1] != 0x00) { printf("bare FF at %zu of %zu\n", i, n); return 1; }
            ++checked;
        }
    }
    printf("ok %ld\n", checked);
    return 0;
}
