/* Plain C user of the C ABI (no torch, no C++): prints the library version and shows the error protocol.
 * Build:  gcc -std=c99 -Iinclude examples/abi_demo.c -Ltexocr_amd -ltexocr_hip -Wl,-rpath,$PWD/texocr_amd -o /tmp/abi_demo
 * A real caller fills txo_config from config/config.yml, hands every state_dict tensor to txo_engine_set_weight, calls
 * txo_engine_finalize_weights and then txo_generate with device pointers. */
#include <stdio.h>
#include <string.h>
#include "texocr.h"

int main(void) {
    printf("%s\n", txo_version());
    txo_config cfg;
    memset(&cfg, 0, sizeof cfg);
    cfg.canvas_h = 224; cfg.canvas_w = 672; cfg.embed = TXO_EMBED_PATCH; cfg.in_channels = 3; cfg.embed_dim = 250; /* not a multiple of 64 */
    cfg.enc_heads = 8; cfg.enc_layers = 4; cfg.dec_heads = 8; cfg.dec_layers = 4; cfg.enc_exp = 4; cfg.dec_exp = 4;
    cfg.vocab = 1000; cfg.max_len = 256; cfg.bos = 998; cfg.eos = 997; cfg.pad = 999; cfg.dtype = TXO_F32; cfg.max_batch = 4;
    txo_engine* e = NULL;
    int rc = txo_engine_create(&cfg, &e);
    printf("create with embed_dim=250 -> %d (%s)\n", rc, txo_last_error());
    return rc == TXO_E_INVALID ? 0 : 1;
}
