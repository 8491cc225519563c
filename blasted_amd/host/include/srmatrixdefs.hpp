// srmatrixdefs.hpp -- reference header name kept for drop-in source compatibility
#pragma once
#include "blasted/storage.hpp"
