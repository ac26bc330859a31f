// tsq_cli -- command-line front end over libturbosqueeze_amd.so, the counterpart of the reference's
// sample program (sample/main.cpp: `tsq c|d|b`), written against include/turbosqueeze.h only.
//
//   tsq_cli c <in> <out.tsq> [--no-ext]     compress a file   (tsqCompress_MT, file -> file)
//   tsq_cli d <in.tsq> <out>                decompress a file (tsqDecompress_MT, file -> file)
//   tsq_cli b [file|--synthetic BYTES] [--no-ext] [--reps N]
//                                           memory -> memory benchmark of the _MT API (host buffers, so
//                                           PCIe transfers are INSIDE the timed region), wall clock, MB = 1e6 B.
// Unlike the reference's benchmark (clock() summed over threads, divided by 1e5) this reports
// wall-clock MB/s.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "turbosqueeze.h"

extern "C" void tsq_synth_text(uint8_t* out, size_t n, uint64_t seed, double s);

static double now_s()
{
    using namespace std::chrono;
    return duration<double>(steady_clock::now().time_since_epoch()).count();
}

static int usage()
{
    fprintf(stderr, "usage: tsq_cli c <in> <out> [--no-ext] | d <in> <out> | b [file | --synthetic BYTES] [--no-ext] [--reps N]\n"
                    "       tsq_cli bf <in> <workdir> [--no-ext] [--reps N]   (file -> file -> file with warm contexts)\n");
    return 2;
}

int main(int argc, char** argv)
{
    if (argc < 2) return usage();
    const std::string mode = argv[1];
    bool ext = true;
    int reps = 3;
    size_t synthetic = 0;
    std::vector<std::string> pos;
    for (int i = 2; i < argc; ++i) {
        if (!strcmp(argv[i], "--no-ext")) ext = false;
        else if (!strcmp(argv[i], "--reps") && i + 1 < argc) reps = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--synthetic") && i + 1 < argc) synthetic = strtoull(argv[++i], nullptr, 10);
        else pos.push_back(argv[i]);
    }
    if (mode == "c" || mode == "d") {
        if (pos.size() != 2) return usage();
        uint8_t* out_path = reinterpret_cast<uint8_t*>(const_cast<char*>(pos[1].c_str()));
        uint8_t* in_path = reinterpret_cast<uint8_t*>(const_cast<char*>(pos[0].c_str()));
        bool ok;
        const double t0 = now_s();
        if (mode == "c") {
            TSQCompressionContext_MT* ctx = tsqAllocateContextCompression_MT(false);
            if (!ctx) { fprintf(stderr, "no usable MI355X (gfx950) device\n"); return 1; }
            ok = tsqCompress_MT(ctx, in_path, 0, true, &out_path, nullptr, true, ext, 0);
            tsqDeallocateContextCompression_MT(ctx);
        } else {
            TSQDecompressionContext_MT* ctx = tsqAllocateContextDecompression_MT(false);
            if (!ctx) { fprintf(stderr, "no usable MI355X (gfx950) device\n"); return 1; }
            ok = tsqDecompress_MT(ctx, in_path, 0, true, &out_path, nullptr, true);
            tsqDeallocateContextDecompression_MT(ctx);
        }
        fprintf(stderr, "%s: %s in %.3f s\n", mode == "c" ? "compress" : "decompress", ok ? "ok" : "FAILED", now_s() - t0);
        return ok ? 0 : 1;
    }
    if (mode == "bf") {
        // file -> .tsq file -> file through the _MT API's file modes with the contexts allocated once (as sample/main.cpp:71-95 keeps
        // context allocation out of its timing): rep 0 warms up (pinned staging, HBM scratch, kernel load), the best of the others counts
        if (pos.size() != 2) return usage();
        const std::string packed = pos[1] + "/bf.tsq", back = pos[1] + "/bf.out";
        uint8_t* in_path = reinterpret_cast<uint8_t*>(const_cast<char*>(pos[0].c_str()));
        uint8_t* packed_path = reinterpret_cast<uint8_t*>(const_cast<char*>(packed.c_str()));
        uint8_t* back_path = reinterpret_cast<uint8_t*>(const_cast<char*>(back.c_str()));
        TSQCompressionContext_MT* cctx = tsqAllocateContextCompression_MT(false);
        TSQDecompressionContext_MT* dctx = tsqAllocateContextDecompression_MT(false);
        if (!cctx || !dctx) { fprintf(stderr, "no usable MI355X (gfx950) device\n"); return 1; }
        double best_c = 1e30, best_d = 1e30;
        for (int r = 0; r <= reps; ++r) {
            remove(packed.c_str()); remove(back.c_str());
            const double t0 = now_s();
            if (!tsqCompress_MT(cctx, in_path, 0, true, &packed_path, nullptr, true, ext, 0)) { fprintf(stderr, "compress failed\n"); return 1; }
            const double t1 = now_s();
            if (!tsqDecompress_MT(dctx, packed_path, 0, true, &back_path, nullptr, true)) { fprintf(stderr, "decompress failed\n"); return 1; }
            const double t2 = now_s();
            if (r > 0) { if (t1 - t0 < best_c) best_c = t1 - t0; if (t2 - t1 < best_d) best_d = t2 - t1; }
        }
        tsqDeallocateContextCompression_MT(cctx);
        tsqDeallocateContextDecompression_MT(dctx);
        auto fsize = [](const std::string& p) -> size_t { FILE* f = fopen(p.c_str(), "rb"); if (!f) return 0; fseek(f, 0, SEEK_END); long n = ftell(f); fclose(f); return n < 0 ? 0 : (size_t)n; };
        const size_t n = fsize(pos[0]), c = fsize(packed);
        // byte-for-byte comparison of the round trip, 64 MiB at a time
        bool exact = n == fsize(back);
        if (exact) {
            FILE* a = fopen(pos[0].c_str(), "rb"); FILE* b = fopen(back.c_str(), "rb");
            std::vector<uint8_t> x(size_t(64) << 20), y(size_t(64) << 20);
            for (size_t got; exact && (got = fread(x.data(), 1, x.size(), a)) > 0;) exact = fread(y.data(), 1, got, b) == got && memcmp(x.data(), y.data(), got) == 0;
            fclose(a); fclose(b);
        }
        printf("{\"mode\": \"file to file\", \"input_bytes\": %zu, \"compressed_bytes\": %zu, \"ext\": %d, \"output_correct\": %s, "
               "\"compress_GBps_wall\": %.2f, \"decompress_GBps_wall\": %.2f, \"compress_s\": %.3f, \"decompress_s\": %.3f, \"reps\": %d}\n",
               n, c, ext ? 1 : 0, exact ? "true" : "false", (double)n / best_c / 1e9, (double)n / best_d / 1e9, best_c, best_d, reps);
        return exact ? 0 : 1;
    }
    if (mode != "b") return usage();

    std::vector<uint8_t> input;
    if (!pos.empty()) {
        FILE* f = fopen(pos[0].c_str(), "rb");
        if (!f) { fprintf(stderr, "cannot open %s\n", pos[0].c_str()); return 1; }
        fseek(f, 0, SEEK_END); long sz = ftell(f); fseek(f, 0, SEEK_SET);
        input.resize((size_t)sz);
        if (fread(input.data(), 1, input.size(), f) != input.size()) { fprintf(stderr, "short read\n"); return 1; }
        fclose(f);
    } else {
        input.resize(synthetic ? synthetic : (size_t)1000000000);
        tsq_synth_text(input.data(), input.size(), 1, 0.0);           // enwik9-shaped text
    }
    TSQCompressionContext_MT* cctx = tsqAllocateContextCompression_MT(false);
    TSQDecompressionContext_MT* dctx = tsqAllocateContextDecompression_MT(false);
    if (!cctx || !dctx) { fprintf(stderr, "no usable MI355X (gfx950) device\n"); return 1; }
    double best_c = 1e30, best_d = 1e30;
    size_t csize = 0;
    bool exact = true;
    for (int r = 0; r <= reps; ++r) {                                  // r == 0 warms up (pinned buffers, HBM scratch)
        uint8_t* comp = nullptr; size_t comp_sz = 0;
        double t0 = now_s();
        if (!tsqCompress_MT(cctx, input.data(), input.size(), false, &comp, &comp_sz, false, ext, 0)) { fprintf(stderr, "compress failed\n"); return 1; }
        double t1 = now_s();
        uint8_t* back = nullptr; size_t back_sz = 0;
        if (!tsqDecompress_MT(dctx, comp, comp_sz, false, &back, &back_sz, false)) { fprintf(stderr, "decompress failed\n"); return 1; }
        double t2 = now_s();
        exact = exact && back_sz == input.size() && memcmp(back, input.data(), back_sz) == 0;
        csize = comp_sz;
        free(comp); free(back);
        if (r > 0) { if (t1 - t0 < best_c) best_c = t1 - t0; if (t2 - t1 < best_d) best_d = t2 - t1; }
    }
    tsqDeallocateContextCompression_MT(cctx);
    tsqDeallocateContextDecompression_MT(dctx);
    printf("{\"input_bytes\": %zu, \"compressed_bytes\": %zu, \"ratio\": %.4f, \"ext\": %d, \"output_correct\": %s, "
           "\"compress_MBps_wall_host_buffers\": %.1f, \"decompress_MBps_wall_host_buffers\": %.1f, \"reps\": %d}\n",
           input.size(), csize, (double)csize / (double)input.size(), ext ? 1 : 0, exact ? "true" : "false",
           (double)input.size() / best_c / 1e6, (double)input.size() / best_d / 1e6, reps);
    return exact ? 0 : 1;
}
