// urdf.cpp -- URDF+ reader (placeholder until the reader lands)
#include <string>
#include <vector>

#include "../../include/grbda_hip.h"

namespace grbda_hip {
int urdf_to_blob(const char *const *, int, int, std::vector<unsigned char> &, std::string &err)
{
    err = "URDF+ reader not built yet";
    return GRBDA_EUNSUPPORTED;
}
}  // namespace grbda_hip
