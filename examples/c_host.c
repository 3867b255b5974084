/* A C99 host on the C ABI (include/i2s.h): no Python, no C++.  Builds with
 *     gcc -std=c99 -Wall -pedantic -I include examples/c_host.c -o c_host -L img2sgf_amd -li2s_hip -Wl,-rpath,$PWD/img2sgf_amd
 * Reads a binary PGM (P5, maxval 255) -- the reference's `input_image_np` for a greyscale scan (img2sgf.py:150) -- runs the board
 * detection on GPU 0 and prints what the reference's "save" button would write (to_SGF, img2sgf.py:781-810): the 19 x 19 matrix as
 * SGF properties.  Without an argument it prints the ABI version and the default parameters and tries to create a context.
 * Exit codes: 0 done, 2 no GPU (i2s_create returned I2S_E_NO_DEVICE: there is no CPU fallback), 1 anything else. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "i2s.h"

static unsigned char* read_pgm(const char* path, int* w, int* h)
{
    FILE* f = fopen(path, "rb");
    int maxval = 0;
    unsigned char* p = NULL;
    if (!f) return NULL;
    /* sizes the library accepts (i2s_create: 1 .. 16384 per side), checked before anything is allocated from the header's numbers */
    if (fscanf(f, "P5 %d %d %d", w, h, &maxval) == 3 && maxval == 255 && *w >= 1 && *h >= 1 && *w <= 16384 && *h <= 16384 && fgetc(f) != EOF) {
        p = (unsigned char*)malloc((size_t)*w * (size_t)*h);
        if (p && fread(p, 1, (size_t)*w * (size_t)*h, f) != (size_t)*w * (size_t)*h) { free(p); p = NULL; }
    }
    fclose(f);
    return p;
}

int main(int argc, char** argv)
{
    i2s_params p;
    i2s_ctx* ctx = NULL;
    int rc, w = 64, h = 64;
    unsigned char* img = NULL;
    i2s_default_params(&p);
    printf("i2s ABI version %d; defaults: Canny %d / %d, HoughCircles (%g, %d, %d, %d, %d), black threshold %d\n", i2s_abi_version(),
           p.canny_lo, p.canny_hi, (double)p.hc_min_dist, p.hc_param1, p.hc_param2, p.hc_min_radius, p.hc_max_radius, p.black_threshold);
    if (argc > 1 && !(img = read_pgm(argv[1], &w, &h))) { fprintf(stderr, "%s: not a binary PGM with maxval 255\n", argv[1]); return 1; }
    rc = i2s_create(&ctx, 0, 1, w, h);
    if (rc == I2S_E_NO_DEVICE) { printf("i2s_create: %s\n", i2s_strerror(rc)); free(img); return 2; }
    if (rc != I2S_OK) { fprintf(stderr, "i2s_create: %s\n", i2s_strerror(rc)); free(img); return 1; }
    if (img) {
        const uint8_t* ptrs[1];
        int ws[1], hs[1], strides[1], chans[1];
        i2s_board board;
        ptrs[0] = img; ws[0] = w; hs[0] = h; strides[0] = w; chans[0] = 1;
        rc = i2s_detect_batch(ctx, 1, ptrs, ws, hs, strides, chans, &p, &board, NULL);
        if (rc != I2S_OK) { fprintf(stderr, "i2s_detect_batch: %s (%s)\n", i2s_strerror(rc), i2s_last_error(ctx)); i2s_destroy(ctx); free(img); return 1; }
        if (board.status != 0) printf("board not detected (status %d)\n", (int)board.status);
        else {
            /* to_SGF, img2sgf.py:781-810, byte for byte: the side to move's stones first; a line per colour even when it has no stone */
            int k, i, j;
            printf("(;GM[1]FF[4]SZ[19]\nPL[%s]\n", board.side_to_move == 1 ? "B" : "W");
            for (k = 0; k < 2; k++) {
                const int colour = (board.side_to_move == 1) == (k == 0) ? I2S_BLACK : I2S_WHITE;
                int any = 0;
                for (i = 0; i < I2S_BOARD_SIZE; i++)
                    for (j = 0; j < I2S_BOARD_SIZE; j++)
                        if (board.board[i][j] == colour) {
                            if (!any) printf(colour == I2S_WHITE ? "AW" : "AB");
                            any = 1;
                            printf("[%c%c]", 'a' + i, 'a' + j);
                        }
                printf("\n");
            }
            printf(")\n");
        }
    }
    i2s_destroy(ctx);
    free(img);
    return 0;
}
