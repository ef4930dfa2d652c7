/* Torch-free host program on the C ABI (include/texocr.h): builds an engine from a flat file written by
 * examples/make_tiny_blob.py (the tiny fixture captured from the reference), runs txo_generate on the GPU and compares the
 * greedy tokens with the reference's.  Plain C + the HIP runtime API for device memory; no Python, no torch.
 *
 * Build (what __graft_entry__.build() does):
 *   hipcc -x c -std=c99 -Iinclude examples/generate_tiny.c -Ltexocr_amd -ltexocr_hip -Wl,-rpath,$PWD/texocr_amd -o examples/generate_tiny
 * Run:   python examples/make_tiny_blob.py /tmp/tiny.bin && examples/generate_tiny /tmp/tiny.bin
 */
#define __HIP_PLATFORM_AMD__ 1
#include <hip/hip_runtime_api.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "texocr.h"

static void die(const char* what) { fprintf(stderr, "generate_tiny: %s (%s)\n", what, txo_last_error()); exit(2); }
static void rd(void* p, size_t n, FILE* f) { if (fread(p, 1, n, f) != n) { fprintf(stderr, "generate_tiny: short read\n"); exit(2); } }

int main(int argc, char** argv) {
    if (argc < 2) { fprintf(stderr, "usage: %s BLOB\n", argv[0]); return 2; }
    FILE* f = fopen(argv[1], "rb");
    if (!f) { perror(argv[1]); return 2; }
    char magic[4];
    rd(magic, 4, f);
    if (memcmp(magic, "TXOB", 4)) { fprintf(stderr, "not a TXOB file\n"); return 2; }
    txo_config cfg;                                      /* 19 int32 fields, written in declaration order */
    rd(&cfg, sizeof cfg, f);
    txo_engine* e = NULL;
    if (txo_engine_create(&cfg, &e)) die("txo_engine_create");
    int32_t nt;
    rd(&nt, 4, f);
    for (int i = 0; i < nt; ++i) {
        int32_t klen, ndim;
        char key[256];
        int64_t shape[4];
        rd(&klen, 4, f);
        if (klen <= 0 || klen >= (int)sizeof key) { fprintf(stderr, "bad key length\n"); return 2; }
        rd(key, (size_t)klen, f); key[klen] = 0;
        rd(&ndim, 4, f);
        if (ndim < 1 || ndim > 4) { fprintf(stderr, "bad rank\n"); return 2; }
        rd(shape, 8 * (size_t)ndim, f);
        size_t n = 1;
        for (int k = 0; k < ndim; ++k) n *= (size_t)shape[k];
        float* w = (float*)malloc(n * 4);
        rd(w, n * 4, f);
        if (txo_engine_set_weight(e, key, w, shape, ndim)) die(key);       /* = load_state_dict, one reference key at a time */
        free(w);
    }
    if (txo_engine_finalize_weights(e)) die("txo_engine_finalize_weights");
    int32_t dims[4], ml[2];
    rd(dims, 16, f);
    const size_t npix = (size_t)dims[0] * dims[1] * dims[2] * dims[3];
    float* img = (float*)malloc(npix * 4);
    rd(img, npix * 4, f);
    rd(ml, 8, f);
    const int B = dims[0], max_len = ml[0], want_steps = ml[1];
    int64_t* want = (int64_t*)malloc((size_t)B * want_steps * 8);
    rd(want, (size_t)B * want_steps * 8, f);
    fclose(f);

    float* d_img = NULL; int64_t* d_tok = NULL;
    if (hipMalloc((void**)&d_img, npix * 4) != hipSuccess || hipMalloc((void**)&d_tok, (size_t)B * max_len * 8) != hipSuccess) die("hipMalloc");
    if (hipMemcpy(d_img, img, npix * 4, hipMemcpyHostToDevice) != hipSuccess) die("hipMemcpy");
    int32_t n_steps = 0;
    /* OCRModel.generate (model/ocr_model.py:46-66): encode, BOS, greedy loop with the GLOBAL eos break; NULL stream = default */
    if (txo_generate(e, d_img, B, dims[1], dims[2], dims[3], max_len, cfg.eos, d_tok, &n_steps, NULL, NULL)) die("txo_generate");
    int64_t* got = (int64_t*)malloc((size_t)B * max_len * 8);
    if (hipMemcpy(got, d_tok, (size_t)B * max_len * 8, hipMemcpyDeviceToHost) != hipSuccess) die("hipMemcpy back");
    int bad = n_steps != want_steps;
    for (int b = 0; b < B && !bad; ++b)
        for (int t = 0; t < want_steps; ++t)
            if (got[(size_t)b * max_len + t] != want[(size_t)b * want_steps + t]) { bad = 1; fprintf(stderr, "row %d step %d: got %lld, reference %lld\n", b, t, (long long)got[(size_t)b * max_len + t], (long long)want[(size_t)b * want_steps + t]); break; }
    printf("%s: %d images, %d steps (reference %d): tokens %s\n", txo_version(), B, n_steps, want_steps, bad ? "DIFFER" : "match the reference");
    /* per-row stop (build extension, texocr.h: txo_set_stop_mode): the same call returns the same steps, every row equal to the reference up to
     * its first eos and cfg.pad behind it */
    if (!bad && cfg.eos >= 0) {
        int32_t n2 = 0;
        if (txo_set_stop_mode(e, TXO_STOP_ROW)) die("txo_set_stop_mode");
        if (txo_generate(e, d_img, B, dims[1], dims[2], dims[3], max_len, cfg.eos, d_tok, &n2, NULL, NULL)) die("txo_generate (row stop)");
        if (txo_set_stop_mode(e, TXO_STOP_GLOBAL)) die("txo_set_stop_mode");
        if (hipMemcpy(got, d_tok, (size_t)B * max_len * 8, hipMemcpyDeviceToHost) != hipSuccess) die("hipMemcpy back");
        bad = n2 != want_steps;
        for (int b = 0; b < B && !bad; ++b) {
            int done = cfg.bos == cfg.eos;
            for (int t = 0; t < want_steps && !bad; ++t) {
                const int64_t w = done ? cfg.pad : want[(size_t)b * want_steps + t];
                if (got[(size_t)b * max_len + t] != w) bad = 1;
                if (want[(size_t)b * want_steps + t] == cfg.eos) done = 1;
            }
        }
        printf("per-row stop: %d steps, tokens %s\n", n2, bad ? "DIFFER" : "match the reference up to each row's eos, pad behind it");
    }
    txo_engine_destroy(e);
    hipFree(d_img); hipFree(d_tok);
    free(img); free(want); free(got);
    return bad;
}
