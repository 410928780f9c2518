/* oracle/oracle_abi.h -- shared structs of the CPU oracle (TEST INFRASTRUCTURE, see mapfill_oracle.c) */
#ifndef ORACLE_ABI_H
#define ORACLE_ABI_H
#include <stdint.h>

#define KIND_GATE 0
#define KIND_RHO 1
#define KIND_EFFECT 2

typedef struct {
    int32_t D;            /* state dimension d^2 */
    int32_t n_rows;       /* prefix-table rows == number of expanded circuits */
    int32_t cache_size;   /* number of cached prefix states */
    const int32_t *t_dest, *t_start, *t_cache, *t_rho; /* per row; -1 == None */
    const int64_t *row_ptr;   /* [n_rows+1] into gate_idx */
    const int32_t *gate_idx;  /* gate applications of each row */
    const int64_t *eff_ptr;   /* [n_rows+1], indexed by t_dest (expanded-circuit index) */
    const int32_t *eff_label; /* effect index per element */
    const int32_t *eff_dest;  /* destination element index per element */
} oracle_plan;

typedef struct {
    int32_t nG, nR, nEl;
    const double *gates, *rhos, *effects;     /* base model, C-contiguous */
    int32_t nP;
    const int32_t *pkind, *pobj, *pelem;      /* parameter p -> (kind, object, flat element) */
} oracle_model;

#endif
