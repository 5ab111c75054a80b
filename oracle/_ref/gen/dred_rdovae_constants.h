/* compile-only stub */
#ifndef DRED_RDOVAE_CONSTANTS_H
#define DRED_RDOVAE_CONSTANTS_H
#define DRED_MAX_RNN_NEURONS 16
#define DRED_MAX_CONV_INPUTS 16
#endif
