// The matrix-pipe form with one series split over the four waves of a workgroup (fista_mfma4.h): A blocks of 32 samples
// per wave, 96 A < N <= 128 A; HRFs of up to 33 taps.
#include "fista_mfma4.h"
#if !defined(PB_A)
#error "compile with -DPB_A=<blocks per wave>"
#endif
namespace pb {
template int launch_mfma4<PB_A>(const FistaArgs&, const double*, int, bool, hipStream_t);
}
