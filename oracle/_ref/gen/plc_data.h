/* compile-only stub: plc_data.h is a generated file absent from the reference tree */
#ifndef PLC_DATA_H
#define PLC_DATA_H
#include "nnet.h"
#define PLC_DENSE1_OUT_SIZE 128
#define PLC_GRU1_STATE_SIZE 16
#define PLC_GRU2_STATE_SIZE 16
#define PLC_MAX_RNN_NEURONS 16
typedef struct { DenseLayer plc_dense1; GRULayer plc_gru1; GRULayer plc_gru2; DenseLayer plc_out; } PLCModel;
typedef struct { float plc_gru1_state[PLC_GRU1_STATE_SIZE]; float plc_gru2_state[PLC_GRU2_STATE_SIZE]; } PLCNetState;
int init_plc_model(PLCModel *model, const WeightArray *arrays);
#endif
