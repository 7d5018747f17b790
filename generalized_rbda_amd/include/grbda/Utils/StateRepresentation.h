// Utils/StateRepresentation.h -- same include path as the reference (include/grbda/Utils/StateRepresentation.h); the facade lives in grbda.h
#pragma once
#include "../grbda.h"
