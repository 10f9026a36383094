// Pre-include for the oracle/_ref build (test infrastructure): pulls in the standard headers BEFORE
// the Makefile's -Dexception=runtime_error takes effect, so that only the reference's MSVC-only
// `throw std::exception("...")` (common/floatimage/floatimage.cpp:272) is retargeted.
// Nothing of the reference is copied or replaced: its sources are compiled where they lie.
#pragma once
#undef exception
#include <algorithm>
#include <cassert>
#include <cmath>
#include <cstring>
#include <exception>
#include <fstream>
#include <iostream>
#include <memory>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>
#include <malloc.h>
#include <omp.h>
#include <xmmintrin.h>
#define GLM_ENABLE_EXPERIMENTAL
#define GLM_FORCE_SIZE_T_LENGTH
#include <glm/glm.hpp>
#include <glm/ext.hpp>
#include <glm/gtx/norm.hpp>
#define exception runtime_error
