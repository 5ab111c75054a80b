/* oracle/refgen.py: symbols the reference links against that live in absent generated files */
#include "nnet.h"
#include "plc_data.h"
int init_plc_model(PLCModel *model, const WeightArray *arrays) { (void)model; (void)arrays; return 1; }
/* ceps_codebooks.c is absent: storage only, filled at run time through ref_set_codebooks(). */
float ceps_codebook1[1024*17];
float ceps_codebook2[1024*17];
float ceps_codebook3[1024*17];
float ceps_codebook_diff4[4096*18];
