// One (PB_S, PB_KT) specialisation of the register-resident FISTA kernel.
#include "launch_fast.h"
#ifndef PB_S
#error "compile with -DPB_S=<samples per lane> -DPB_KT=<taps>"
#endif
namespace pb {
template int launch_fast<PB_S, PB_KT>(const FistaArgs&, const double*, int, bool, int, hipStream_t);
template int launch_fast_pp<PB_S, PB_KT>(const FistaArgs&, int, hipStream_t);

}
