// Dynamics/ClusterTreeModel.h -- same include path as the reference (include/grbda/Dynamics/ClusterTreeModel.h); the facade lives in grbda.h
#pragma once
#include "../grbda.h"
