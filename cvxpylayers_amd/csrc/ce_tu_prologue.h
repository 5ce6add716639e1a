// common prologue of every kernel translation unit
#pragma once
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <string>
#include <type_traits>
#include <utility>
#include <vector>

#include "cone_engine.h"
#include "ce_types.h"
