// Dynamics/ClusterJoints/LoopConstraint.h -- same include path as the reference (include/grbda/Dynamics/ClusterJoints/LoopConstraint.h); the facade lives in grbda.h
#pragma once
#include "../../grbda.h"
