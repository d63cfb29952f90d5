// CSR SpMM launcher of csrc/lightgcn.hip, shared with NGCF's propagation (ngcf.hip).
#pragma once
#include "common.hpp"

namespace hiprec {

// y = (A with dropped edges) x ; acc += y (acc may be NULL).  The kernel accumulates the slices of heavy
// rows with atomics, so y must be zero: y_is_zero = true when the caller cleared it already (several
// outputs in one memset), false to clear it here.
int launch_spmm(const hiprec_csr* a, const uint8_t* keep, float scale, const float* x, float* y, float* acc,
                int dim, hipStream_t st, bool y_is_zero = false);

}  // namespace hiprec
