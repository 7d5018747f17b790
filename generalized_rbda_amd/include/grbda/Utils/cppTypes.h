// Utils/cppTypes.h -- same include path as the reference (include/grbda/Utils/cppTypes.h); the facade lives in grbda.h
#pragma once
#include "../grbda.h"
