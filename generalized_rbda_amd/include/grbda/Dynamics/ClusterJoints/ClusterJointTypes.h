// Dynamics/ClusterJoints/ClusterJointTypes.h -- same include path as the reference (include/grbda/Dynamics/ClusterJoints/ClusterJointTypes.h); the facade lives in grbda.h
#pragma once
#include "../../grbda.h"
