// One (PB_S, PB_KT) specialisation of the all-float64 register-resident kernel.
#include "fista_exact.h"
#ifndef PB_S
#error "compile with -DPB_S=<samples per lane> -DPB_KT=<taps>"
#endif
namespace pb {
template int launch_exact<PB_S, PB_KT>(const FistaArgs&, const double*, int, bool, int, hipStream_t);
}
