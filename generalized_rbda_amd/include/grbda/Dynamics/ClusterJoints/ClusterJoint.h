// Dynamics/ClusterJoints/ClusterJoint.h -- same include path as the reference (include/grbda/Dynamics/ClusterJoints/ClusterJoint.h); the facade lives in grbda.h
#pragma once
#include "../../grbda.h"
