// Mutation fuzzer for the two file formats next to the scan path: the HMMER3 ASCII reader
// (dcp_h3reader_*) and the dcpx profile DB (dcp_db_*).  Built with -fsanitize=address,undefined
// by tests/test_sanitizers.py: every mutated file must end in DCP_OK / DCP_END or a clean error
// code -- never a crash, an out-of-bounds access or a leak.
//   fuzz_parsers good.hmm good.dcpx scratch_dir iterations seed
#include "dcp_gpu.h"
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

static uint64_t rng_state;
static uint64_t rnd()
{
    rng_state ^= rng_state << 13, rng_state ^= rng_state >> 7, rng_state ^= rng_state << 17;
    return rng_state;
}

static std::vector<unsigned char> slurp(char const *path)
{
    std::vector<unsigned char> b;
    if (FILE *f = std::fopen(path, "rb"))
    {
        unsigned char buf[4096];
        size_t n;
        while ((n = std::fread(buf, 1, sizeof buf, f)) > 0)
            b.insert(b.end(), buf, buf + n);
        std::fclose(f);
    }
    return b;
}

static void mutate(std::vector<unsigned char> &b, bool text)
{
    unsigned const nmut = 1 + (unsigned)(rnd() % 4);
    for (unsigned m = 0; m < nmut && !b.empty(); ++m)
    {
        size_t const at = rnd() % b.size();
        switch (rnd() % 5)
        {
        case 0: b[at] = text ? (unsigned char)" \n*.-0123456789eEHMLN/"[rnd() % 22] : (unsigned char)rnd(); break;
        case 1: b.resize(at); break;                                         // truncate
        case 2: b.insert(b.begin() + at, b.begin() + at, b.begin() + std::min(b.size(), at + 1 + rnd() % 64)); break; // duplicate a run
        case 3: b.erase(b.begin() + at, b.begin() + std::min(b.size(), at + 1 + rnd() % 64)); break;
        default: b[at] ^= (unsigned char)(1u << (rnd() % 8)); break;          // bit flip
        }
    }
}

static bool dump(std::string const &path, std::vector<unsigned char> const &b)
{
    FILE *f = std::fopen(path.c_str(), "wb");
    if (!f) return false;
    if (!b.empty()) std::fwrite(b.data(), 1, b.size(), f);
    std::fclose(f);
    return true;
}

int main(int argc, char **argv)
{
    if (argc < 6) return 2;
    std::vector<unsigned char> const hmm = slurp(argv[1]), dbf = slurp(argv[2]);
    std::string const dir = argv[3];
    unsigned const iters = (unsigned)std::atoi(argv[4]);
    rng_state = 0x9E3779B97F4A7C15ull ^ (uint64_t)std::atoll(argv[5]);
    if (hmm.empty() || dbf.empty()) return 2;
    unsigned ok_hmm = 0, ok_db = 0;
    for (unsigned it = 0; it < iters; ++it)
    {
        {
            std::vector<unsigned char> b = hmm;
            mutate(b, true);
            std::string const p = dir + "/m.hmm";
            if (!dump(p, b)) return 3;
            dcp_h3reader *r = dcp_h3reader_open(p.c_str(), DCP_ENTRY_DIST_OCCUPANCY, 0.01f);
            if (r)
            {
                dcp_profile *prof = nullptr;
                int rc;
                unsigned n = 0;
                while ((rc = dcp_h3reader_next(r, &prof)) == DCP_OK && n < 64)
                {
                    if (!prof || dcp_profile_core_size(prof) == 0) return 4;
                    dcp_profile_del(prof);
                    ++n;
                }
                if (rc == DCP_END) ++ok_hmm;
                else if (rc != DCP_OK && !dcp_h3reader_error(r)) return 5; // an error must carry a message
                dcp_h3reader_close(r);
            }
        }
        {
            std::vector<unsigned char> b = dbf;
            mutate(b, false);
            std::string const p = dir + "/m.dcpx";
            if (!dump(p, b)) return 3;
            int rc = -1;
            dcp_db *db = dcp_db_open(p.c_str(), &rc);
            if (db)
            {
                unsigned const n = dcp_db_nprofiles(db);
                uint32_t const *sz = dcp_db_profile_sizes(db);
                for (unsigned i = 0; i < n; ++i)
                    if (sz[i] == 0) return 7;
                unsigned psize[DCP_NUM_THREADS];
                int64_t poff[DCP_NUM_THREADS + 1];
                (void)dcp_db_partitions(db, 1 + (unsigned)(rnd() % DCP_NUM_THREADS), psize, poff);
                std::vector<dcp_profile *> out(n, nullptr);
                if (n && dcp_db_read(db, 0, n, out.data()) == DCP_OK) ++ok_db;
                for (dcp_profile *q : out)
                    dcp_profile_del(q);
                dcp_db_close(db);
            }
            else if (rc == DCP_OK) return 6; // a failed open must say why
        }
    }
    // the model builder with hostile parameters: -inf / NaN / huge log-probabilities, edge core sizes,
    // bad entry_dist / epsilon, missing consensus -- a profile or DCP_EINVAL, and a finite-or--inf table
    unsigned built = 0;
    for (unsigned it = 0; it < iters / 8 + 8; ++it)
    {
        static unsigned const sizes[] = {0, 1, 2, 3, 64, 65, 257, 4096, 4097};
        unsigned const M = sizes[rnd() % 9];
        size_t const Mm = M ? M : 1;
        auto val = [&]() -> float {
            switch (rnd() % 12)
            {
            case 0: return -__builtin_inff();
            case 1: return __builtin_nanf("");
            case 2: return 1e30f;
            case 3: return -1e30f;
            case 4: return 0.0f;
            default: return -(float)(rnd() % 100000) / 10000.0f;
            }
        };
        std::vector<float> nul(20), mat(Mm * 20), tr((Mm + 1) * 7);
        for (float &v : nul) v = val();
        for (float &v : mat) v = val();
        for (float &v : tr) v = val();
        int const entry = (int)(rnd() % 5) - 1;
        float const eps = (rnd() % 4 == 0) ? val() : 0.01f;
        std::string const cons(rnd() % 3 ? Mm : Mm / 2, 'a');
        int rc = -77;
        dcp_profile *p = dcp_profile_new(rnd() % 5 ? "acc" : nullptr, M, entry, eps, nul.data(), mat.data(), tr.data(),
                                         rnd() % 4 ? cons.c_str() : nullptr, &rc);
        if (!p && rc == DCP_OK) return 8;
        if (p)
        {
            if (M == 0 || M > 4096) return 9;
            float tab[DCP_NCODES];
            dcp_frame_table_host(dcp_profile_match_dist(p) + (size_t)(M - 1) * DCP_NDIST, 0.01f, tab);
            ++built;
            dcp_profile_del(p);
        }
    }
    // decode and product rows with hostile inputs: symbols outside ACGT, fragment lengths 0..8, state ids
    // of nodes the profile does not have, steps that run past the sequence, tiny output buffers
    unsigned rows = 0;
    {
        int rc = -1;
        dcp_profile *p = dcp_profile_sample("acc", 11, 9, DCP_ENTRY_DIST_OCCUPANCY, 0.01f, &rc);
        if (!p) return 10;
        for (unsigned it = 0; it < iters; ++it)
        {
            uint8_t seq[40], codon[3];
            unsigned const L = (unsigned)(rnd() % 41);
            for (unsigned i = 0; i < L; ++i)
                seq[i] = (uint8_t)(rnd() % 16 ? rnd() % 4 : rnd());
            unsigned const sid = (unsigned)(rnd() % 8 ? ((rnd() % 4) << 14) | (rnd() % 12) : rnd() & 0xffff);
            (void)dcp_profile_decode(p, seq, (unsigned)(rnd() % 9), sid, codon);
            dcp_step steps[24];
            unsigned const ns = (unsigned)(rnd() % 25);
            for (unsigned i = 0; i < ns; ++i)
                steps[i] = dcp_step{(uint16_t)(rnd() % 6 ? ((rnd() % 4) << 14) | (rnd() % 11) : rnd()), (uint8_t)(rnd() % 7), 0};
            char row[600];
            size_t const cap = rnd() % 5 ? sizeof row : (size_t)(rnd() % 64);
            long const n = dcp_prod_format_row(row, cap, (int64_t)rnd(), -3, rnd() % 7 ? "PF00001.1" : "", "dna", -1.5, -2.5,
                                               "protein", "0.1.0", p, seq, L, steps, ns);
            if (n > (long)cap) return 11;
            if (n > 0)
            {
                if (row[n - 1] != '\n') return 12;
                ++rows;
            }
        }
        dcp_profile_del(p);
    }
    std::printf("fuzz_parsers ok: %u iterations, %u hmm and %u dcpx mutants still parsed, %u hostile profiles built, %u rows\n", iters,
                ok_hmm, ok_db, built, rows);
    return 0;
}
