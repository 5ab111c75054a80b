/* lpcnet_hip_state.h -- layout of `struct LPCNetState` of the HIP engine, spelled with the REFERENCE's member names.
 *
 * The reference's packet-loss concealment (src/lpcnet_plc.c) is the one caller that looks inside the state: it embeds
 * LPCNetState by value in LPCNetPLCState (src/lpcnet_private.h:86), copies it (snapshot / rollback, src/lpcnet_plc.c:223-230,
 * :384-414) and clears five members directly (src/lpcnet_plc.c:176-180):
 *     st->lpcnet.last_sig, .last_exc, .deemph_mem, .nnet.gru_a_state, .nnet.gru_b_state
 * The engine's state is a relocatable POD as well, so it can give those members the reference's NAMES (at the engine's own
 * offsets): with this definition the unmodified src/lpcnet_plc.c compiles and links against liblpcnet_hip.so
 * (integration/lpcnet_private_hip.h, integration/Makefile target plc, tests/test_demo_integration.py).
 *
 * api.c defines the state it really uses and statically asserts that every member below sits where this header says.
 * Members the engine adds are prefixed hip_; the sizes are the fixed architecture of the default model
 * (the reference's generated nnet_data.h: GRU_A_STATE_SIZE 384, GRU_B_STATE_SIZE 16, FEATURE_CONV1_STATE_SIZE 168,
 * FEATURE_CONV2_STATE_SIZE 256, FEATURES_DELAY 2; src/freq.h: LPC_ORDER 16; include/lpcnet.h: NB_FEATURES 20). */
#ifndef LPCNET_HIP_STATE_H_
#define LPCNET_HIP_STATE_H_

struct LPCNetState {
    unsigned int hip_magic;
    int hip_model_id;                         /* registry slot of the bound model, -1 = none (no pointers in the state) */
    struct {                                  /* reference: NNetState nnet (generated nnet_data.h), engine order */
        float gru_a_state[384];
        float gru_b_state[16];
        float feature_conv1_state[168];
        float feature_conv2_state[256];
    } nnet;
    float old_lpc[2][16];                     /* src/lpcnet_private.h:40 */
    float last_sig[16];                       /* :35 */
    float deemph_mem;                         /* :46 */
    int last_exc;                             /* :34 */
    int frame_count;                          /* :45 */
    unsigned int hip_rng[4];                  /* kiss99 state (reference: kiss99_ctx rng, :31) */
    float lpc[16];                            /* :47 */
    int hip_pad[3];
    float gru_a_condition[3 * 384];           /* :43 */
    float gru_b_condition[3 * 16];            /* :44 */
    float feature_buffer[20 * 4];             /* :36, run_frame_network_deferred queue */
    int feature_buffer_fill;                  /* :37 */
};

#endif
