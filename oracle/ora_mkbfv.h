/* ora_mkbfv.h -- CPU ORACLE (test infrastructure, NOT the product path).
 * Restates mkbfv/{basis_extension,keyswitch,keyswitch_hoisted,evaluator}.go.  PARITY UNPINNED vs Go. */
#ifndef ORA_MKBFV_H
#define ORA_MKBFV_H
#include "ora_mkrlwe.h"
#ifdef __cplusplus
extern "C" {
#endif
/* filled in with the BFV row of SURVEY.md 8(a10) */
#ifdef __cplusplus
}
#endif
#endif
