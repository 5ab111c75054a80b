/* Link-time stand-ins for the entry points the reference's demo (src/lpcnet_demo.c) references for the modes that are
 * OUTSIDE the HIP engine's scope: -encode / -features (feature extraction, src/lpcnet_enc.c), -plc (packet-loss
 * concealment network, src/lpcnet_plc.c) and -addlpc (a feature-file utility built on lpc_from_cepstrum, src/freq.c:310).
 * liblpcnet_hip.so implements the synthesis path (-synthesis, -decode); selecting one of the other modes on a demo
 * linked against it stops with a message instead of silently doing nothing.
 *
 * Written against the reference's own include/lpcnet.h (:103-155 encoder, :197-215 PLC) and src/freq.h:57. */
#include <stdio.h>
#include <stdlib.h>
#include "lpcnet.h"

static void out_of_scope(const char *what)
{
    fprintf(stderr, "%s: this mode is not part of the LPCNet HIP engine (synthesis / decode only); "
                    "link the reference's own liblpcnet for it\n", what);
    exit(2);
}

LPCNetEncState *lpcnet_encoder_create(void) { out_of_scope("lpcnet_encoder_create"); return NULL; }
void lpcnet_encoder_destroy(LPCNetEncState *st) { (void)st; }
int lpcnet_encode(LPCNetEncState *st, const short *pcm, unsigned char *buf) { (void)st; (void)pcm; (void)buf; out_of_scope("lpcnet_encode"); return -1; }
int lpcnet_compute_single_frame_features(LPCNetEncState *st, const short *pcm, float features[NB_TOTAL_FEATURES])
{
    (void)st; (void)pcm; (void)features;
    out_of_scope("lpcnet_compute_single_frame_features");
    return -1;
}

LPCNetPLCState *lpcnet_plc_create(int options) { (void)options; out_of_scope("lpcnet_plc_create"); return NULL; }
void lpcnet_plc_destroy(LPCNetPLCState *st) { (void)st; }
int lpcnet_plc_load_model(LPCNetPLCState *st, const unsigned char *data, int len) { (void)st; (void)data; (void)len; out_of_scope("lpcnet_plc_load_model"); return -1; }
int lpcnet_plc_update(LPCNetPLCState *st, short *pcm) { (void)st; (void)pcm; out_of_scope("lpcnet_plc_update"); return -1; }
int lpcnet_plc_conceal(LPCNetPLCState *st, short *pcm) { (void)st; (void)pcm; out_of_scope("lpcnet_plc_conceal"); return -1; }

float lpc_from_cepstrum(float *lpc, const float *cepstrum) { (void)lpc; (void)cepstrum; out_of_scope("lpc_from_cepstrum (-addlpc)"); return 0.f; }
