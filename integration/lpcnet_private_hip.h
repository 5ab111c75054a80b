/* Forced include for building the reference's src/lpcnet_plc.c UNMODIFIED against the HIP engine:
 *
 *     gcc -include integration/lpcnet_private_hip.h -I<generated model headers> -I$REF/include -I$REF/src -c $REF/src/lpcnet_plc.c
 *
 * src/lpcnet_plc.c includes "lpcnet_private.h", which defines struct LPCNetState with the reference's CPU layout and the
 * structs that embed it (LPCNetDecState, LPCNetPLCState).  This header is seen first: it defines struct LPCNetState with the
 * ENGINE's layout (include/lpcnet_hip_state.h: same member names for what the PLC touches), then pulls in the reference's
 * header itself with the tag of ITS definition renamed, so that LPCNetEncState, LPCNetPLCState (now embedding the engine's
 * state) and the prototypes come from the reference, unedited.  The reference header's include guard is then set, and
 * the #include "lpcnet_private.h" inside lpcnet_plc.c expands to nothing.
 *
 * How the rename works: C keeps struct tags and typedef names apart.  With  #define LPCNetState LPCNetHipShadow  the
 * reference header defines the TAG  struct LPCNetHipShadow {...}  (its CPU layout: unused), while every USE in it goes
 * through the typedef name -- `LPCNetState lpcnet;` becomes `LPCNetHipShadow lpcnet;` -- which we declare as a typedef of
 * the engine's struct. */
#ifndef LPCNET_PRIVATE_HIP_H_
#define LPCNET_PRIVATE_HIP_H_

#include "lpcnet.h"                          /* the REFERENCE's public header (-I$REF/include; not this repository's include/): typedef struct LPCNetState LPCNetState; */
#include "../include/lpcnet_hip_state.h"     /* struct LPCNetState { engine layout, reference member names } */

typedef struct LPCNetState LPCNetHipShadow;
#define LPCNetState LPCNetHipShadow
#include "lpcnet_private.h"                  /* the reference's src/lpcnet_private.h (found through -I$REF/src) */
#undef LPCNetState

#endif
