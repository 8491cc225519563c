// blasted_petsc_ext.hpp -- C++ extension of the PCSHELL surface: install BLASTed with a user factory.
// The reference declares a reference-taking overload (include/blasted_petsc_ext.hpp:27) but defines a
// pointer-taking one (src/blasted_petsc.cpp:578); both are provided here.
#pragma once

#include "blasted_petsc.h"
#include "solverfactory.hpp"

namespace blasted {
int setup_blasted_stack_ext(KSP ksp, const FactoryBase<double, int> &factory, Blasted_data_list *const bctx);
}
int setup_blasted_stack_ext(KSP ksp, const blasted::FactoryBase<double, int> *const factory,
                            Blasted_data_list *const bctx);
