// One (PB_S, PB_KT) specialisation of the two-problems-per-row kernel with 2-parallel fast
// FIRs.  Separate translation unit: compiled with -mllvm -enable-misched=0 (see Makefile).
#include "fista_pair_ffa.h"
#ifndef PB_S
#error "compile with -DPB_S=<samples per lane> -DPB_KT=<taps>"
#endif
namespace pb {
#if PB_S <= 20 && PB_KT <= 32
template int launch_pair_ffa<PB_S, PB_KT>(const FistaArgs&, const double*, int, bool, hipStream_t);
template int launch_pair_ffa_dev<PB_S, PB_KT>(const FistaArgs&, hipStream_t);
template int launch_pair_ffa_cert<PB_S, PB_KT>(const FistaArgs&, const double*, int, hipStream_t);
template int launch_pair_ffa_split<PB_S, PB_KT>(const FistaArgs&, const double*, int, bool, bool, hipStream_t);
#endif
}
