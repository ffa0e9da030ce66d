/* ora_mkbfv.c -- CPU ORACLE (test infrastructure only). BFV path: see ora_mkbfv.h */
#include "ora_mkbfv.h"
