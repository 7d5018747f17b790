// Dynamics/ClusterJoints/Transmissions.h -- same include path as the reference (include/grbda/Dynamics/ClusterJoints/Transmissions.h); the facade lives in grbda.h
#pragma once
#include "../../grbda.h"
