// Sanitizer driver for the product's host-only code (csrc/dcp_model.cpp): profile builder,
// frame tables, special transitions, partitions, codon decode, product rows, HMMER3 reader.
// Built with -fsanitize=address,undefined by tests/test_sanitizers.py (no GPU code involved).
#include "dcp_gpu.h"
#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>

static int failed;
#define CHECK(c)                                                                  \
    do                                                                            \
    {                                                                             \
        if (!(c)) { std::fprintf(stderr, "asan_host_model:%d: %s\n", __LINE__, #c); failed++; } \
    } while (0)

int main(int argc, char **argv)
{
    int rc = -1;
    for (unsigned M : {2u, 7u, 64u, 300u})
    {
        dcp_profile *p = dcp_profile_sample("acc", 5 + M, M, DCP_ENTRY_DIST_OCCUPANCY, 0.01f, &rc);
        CHECK(p && rc == DCP_OK && dcp_profile_core_size(p) == M);
        float tab[DCP_NCODES];
        dcp_frame_table_host(dcp_profile_match_dist(p) + (size_t)(M - 1) * DCP_NDIST, 0.01f, tab);
        double mass = 0;
        for (float v : tab)
            mass += std::exp((double)v);
        CHECK(std::fabs(mass - 1.0) < 1e-5);
        uint8_t frag[5] = {0, 1, 2, 3, 0}, codon[3];
        for (unsigned len = 1; len <= 5; ++len)
            CHECK(dcp_profile_decode(p, frag, len, 1, codon) == DCP_OK && codon[0] < 4);
        CHECK(dcp_profile_decode(p, frag, 3, (3u << 14) | 1u, codon) == DCP_EINVAL); // S is mute
        CHECK(dcp_profile_decode(p, frag, 6, 1, codon) == DCP_EINVAL);
        dcp_step steps[4] = {{(uint16_t)((3u << 14) | 1u), 0, 0}, {(uint16_t)((3u << 14) | 2u), 3, 0},
                             {1, 2, 0}, {(uint16_t)((3u << 14) | 7u), 0, 0}};
        char row[512];
        long n = dcp_prod_format_row(row, sizeof row, 1, 2, "acc", "dna", -1.5, -2.5, "protein", "0.1.0", p, frag, 5, steps, 4);
        CHECK(n > 0 && row[n - 1] == '\n' && std::strstr(row, ",S,,;ACG,N,") != nullptr);
        CHECK(dcp_prod_format_row(row, 16, 1, 2, "acc", "dna", -1.5, -2.5, "protein", "0.1.0", p, frag, 5, steps, 4) == -1);
        dcp_profile_del(p);
    }
    CHECK(dcp_profile_sample("x", 1, 1, DCP_ENTRY_DIST_UNIFORM, 0.01f, &rc) == nullptr && rc == DCP_EINVAL);
    CHECK(dcp_profile_sample("x", 1, 5000, DCP_ENTRY_DIST_UNIFORM, 0.01f, &rc) == nullptr);
    float xt[DCP_NXTRANS];
    CHECK(dcp_xtrans(0, 1, 0, xt) == DCP_EINVAL && dcp_xtrans(1000, 1, 0, xt) == DCP_OK && xt[0] < 0);
    unsigned sizes[DCP_NUM_THREADS];
    CHECK(dcp_partition_by_count(20000, 8, sizes) == 8 && sizes[7] == 2500);
    CHECK(dcp_partition_by_count(3, 64, sizes) == 3 && dcp_partition_by_count(5, 0, sizes) == 0);
    std::vector<unsigned> cs(1000, 100), pb(9);
    dcp_partition_by_cells(cs.data(), 1000, 8, pb.data());
    CHECK(pb[0] == 0 && pb[8] == 1000 && pb[4] == 500);
    char name[8];
    CHECK(dcp_state_name((2u << 14) | 4096u, name) == 5 && !std::strcmp(name, "D4096"));
    if (argc > 1)
    {
        dcp_h3reader *r = dcp_h3reader_open(argv[1], DCP_ENTRY_DIST_OCCUPANCY, 0.01f);
        CHECK(r != nullptr);
        dcp_profile *p = nullptr;
        int n = 0;
        while (r && (rc = dcp_h3reader_next(r, &p)) == DCP_OK)
        {
            CHECK(dcp_profile_core_size(p) > 0 && std::strlen(dcp_profile_consensus(p)) == dcp_profile_core_size(p));
            dcp_profile_del(p);
            n++;
        }
        CHECK(rc == DCP_END && n == 2);
        dcp_h3reader_close(r);
        if (argc > 2)
        {
            r = dcp_h3reader_open(argv[2], DCP_ENTRY_DIST_OCCUPANCY, 0.01f);
            CHECK(r && dcp_h3reader_next(r, &p) == DCP_EPARSE && p == nullptr);
            dcp_h3reader_close(r);
        }
    }
    if (!failed) std::puts("asan_host_model ok");
    return failed;
}
