// Lane-cooperative d = 4 EKF (chirp LCD / La Scala LCD models), see cgp_coop4.hpp.
#include "cgp_coop4.hpp"
namespace cgp {
int dispatch_filter_coop4(const FilterIO& io, const ModelArgs& ma, hipStream_t st) { return launch_ekf4_coop(io, ma, st); }
}  // namespace cgp
