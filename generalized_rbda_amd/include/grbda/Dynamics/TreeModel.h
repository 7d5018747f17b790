// Dynamics/TreeModel.h -- same include path as the reference (include/grbda/Dynamics/TreeModel.h); the facade lives in grbda.h
#pragma once
#include "../grbda.h"
