// One (PB_S, PB_KT) specialisation of the one-problem-per-wave form of the
// register-resident kernel (series of up to 64 * PB_S scans).
#include "launch_fast.h"
#ifndef PB_S
#error "compile with -DPB_S=<samples per lane> -DPB_KT=<taps>"
#endif
namespace pb {
template int launch_wide<PB_S, PB_KT>(const FistaArgs&, const double*, int, bool, int, hipStream_t);
template int launch_wide_pp<PB_S, PB_KT>(const FistaArgs&, int, hipStream_t);
}
