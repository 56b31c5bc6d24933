// The resident receiver's kernels: the RES instances of demodStream (lorahip_streamkernel.h) for the 16-points-per-lane geometries of
// SF7-10 -- one launch stays on the device across the receiver's steps, which arrive as messages (lorahip_demod_receive, async = 3).
// A translation unit of its own: the instances are as large as the ones of lorahip_stream.hip.
#include "lorahip_streamkernel.h"
#include "lorahip_streamcfg.h"

namespace lorahip {

hipError_t launchStreamResident(const int sf, const StreamArgs &s, hipStream_t stream, unsigned *grid)
{
    switch (sf)
    {
    case 7: return launchStreamResidentCfg<Stream7>(s, stream, grid);
    case 8: return launchStreamResidentCfg<Stream8>(s, stream, grid);
    case 9: return launchStreamResidentCfg<Stream9>(s, stream, grid);
    case 10: return launchStreamResidentCfg<Stream10>(s, stream, grid);
    case 11: case 12: return launchStreamResidentWide(sf, s, stream, grid);
    default: return hipErrorNotSupported;
    }
}

} // namespace lorahip
