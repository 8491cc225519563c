// reference header name (include/coomatrix.hpp)
#pragma once
#include "blasted/mmio.hpp"
