// The matrix-pipe form of the fused FISTA kernel (fista_mfma.h): series of 129..310 scans
// (NB blocks of 31 samples + one sum slot; only the last block may hold padding), HRFs of up to 33 taps (two near
// tiles) or 64 taps (three).
#include "fista_mfma.h"
#ifndef PB_NB
#error "compile with -DPB_NB=<blocks of 31 samples>"
#endif
namespace pb {
template int launch_mfma<PB_NB>(const FistaArgs&, const double*, int, bool, hipStream_t);
}
