// async_initialization_decl.hpp -- reference header name kept for drop-in source compatibility
#pragma once
#include "blasted/types.hpp"
