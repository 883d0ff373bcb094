// The matrix-pipe form with one series split over the two waves of a workgroup (fista_mfma2.h): NBA + NBB blocks of
// 32 samples, 32 (NBA+NBB-1) < N <= 32 (NBA+NBB); HRFs of up to 33 taps; plain solves.
#include "fista_mfma2.h"
#if !defined(PB_NBA) || !defined(PB_NBB)
#error "compile with -DPB_NBA=<blocks of the left wave> -DPB_NBB=<blocks of the right wave>"
#endif
namespace pb {
template int launch_mfma2<PB_NBA, PB_NBB>(const FistaArgs&, const double*, int, bool, hipStream_t);
}
