// Host cost of every scan of a progressive JPEG, through the product's own host decoder (csrc/jpeg_host.h), one thread, best of 5:
// what the refinement passes -- the part of row 8f-4 that stays on host threads -- cost per file.
//   g++ -O2 -std=c++17 -I img2sgf_amd/csrc tools/micro/jpeg_refine_cost.cpp -o /tmp/jpeg_refine_cost && /tmp/jpeg_refine_cost tests/golden/test_images/ex1.jpg ...
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define I2S_HD
#include "jpeg_host.h"
using namespace i2s;

static double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argc, char** argv)
{
    for (int a = 1; a < argc; a++) {
        FILE* fp = fopen(argv[a], "rb");
        if (!fp) { perror(argv[a]); return 1; }
        std::vector<uint8_t> d;
        uint8_t buf[65536];
        size_t n;
        while ((n = fread(buf, 1, sizeof buf, fp)) > 0) d.insert(d.end(), buf, buf + n);
        fclose(fp);
        JpegFile f;
        if (jpg_parse(d.data(), d.size(), &f) != JPG_OK) { fprintf(stderr, "%s: not decodable here\n", argv[a]); return 1; }
        const JpegFrameView fv = jpg_frame_view(f);
        std::vector<std::vector<int16_t>> store(3);
        int16_t* coef[3] = {nullptr, nullptr, nullptr};
        std::vector<double> best(f.scans.size(), 1e30);
        for (int rep = 0; rep < 5; rep++) {
            for (int c = 0; c < f.ncomp; c++) { store[c].assign((size_t)f.c[c].bw * f.c[c].bh * 64, 0); coef[c] = store[c].data(); }
            for (size_t si = 0; si < f.scans.size(); si++) {
                const JpegScan& sc = f.scans[si];
                JpegScanView v;
                v.ns = sc.ns; v.ss = sc.ss; v.se = sc.se; v.ah = sc.ah; v.al = sc.al; v.dri = sc.dri;
                for (int k = 0; k < 3; k++) { v.ci[k] = sc.ci[k]; v.td[k] = sc.td[k]; v.ta[k] = sc.ta[k]; }
                for (int t = 0; t < 4; t++) { v.dc[t] = &sc.dc[t]; v.ac[t] = &sc.ac[t]; }
                v.data = sc.data; v.len = sc.len;
                const double t0 = now_ms();
                if (jpg_decode_scan_view(fv, v, coef) != JPG_OK) { fprintf(stderr, "scan %zu failed\n", si); return 1; }
                const double dt = now_ms() - t0;
                if (dt < best[si]) best[si] = dt;
            }
        }
        double first = 0, refine = 0, seen_refine = 0, dev = 0;
        printf("%s: %d x %d, %d component(s), %zu bytes, %zu scans\n", argv[a], f.X, f.Y, f.ncomp, d.size(), f.scans.size());
        for (size_t si = 0; si < f.scans.size(); si++) {
            const JpegScan& sc = f.scans[si];
            const bool r = sc.ah != 0;
            if (r) seen_refine = 1;
            if (!seen_refine) dev += best[si];
            (r ? refine : first) += best[si];
            printf("  scan %2zu: comps %d  Ss %2d Se %2d Ah %d Al %d  %7zu bytes  %6.3f ms  %s\n", si, sc.ns, sc.ss, sc.se, sc.ah, sc.al, sc.len, best[si],
                   r ? (sc.ss == 0 ? "DC refinement (one bit per block)" : "AC refinement") : (seen_refine ? "first pass behind a refinement pass" : "first pass (device)"));
        }
        printf("  first passes %.3f ms (of which in front of the first refinement pass, i.e. on the device today: %.3f), refinement passes %.3f ms\n", first, dev, refine);
    }
    return 0;
}
