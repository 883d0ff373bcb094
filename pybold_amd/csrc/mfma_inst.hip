// The matrix-pipe form of the fused FISTA kernel (fista_mfma.h): series of up to 320 scans.
#include "fista_mfma.h"
namespace pb {
template int launch_mfma<10>(const FistaArgs&, const double*, int, hipStream_t);
}
