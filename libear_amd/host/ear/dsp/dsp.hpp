// ear/dsp/dsp.hpp — everything of the HIP-backed ear::dsp in one include: the five
// component classes libear's umbrella header (include/ear/dsp/dsp.hpp:2-6) pulls in,
// plus the composed Objects renderer, which is the batched entry point of this library.
#pragma once
#include "objects_renderer.hpp"      // gains -> buses -> decorrelate -> delay -> mix, many blocks per call
#include "gain_interpolator.hpp"     // GainInterpolator<LinearInterp{Single,Vector,Matrix}>
#include "block_convolver.hpp"       // block_convolver::{Context, Filter, BlockConvolver}
#include "delay_buffer.hpp"          // DelayBuffer
#include "variable_block_size.hpp"   // VariableBlockSizeAdapter
#include "ptr_adapter.hpp"           // PtrAdapter / PtrAdapterConst
