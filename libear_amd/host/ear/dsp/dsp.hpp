// ear/dsp/dsp.hpp — umbrella header, like libear's include/ear/dsp/dsp.hpp:2-6
#pragma once
#include "block_convolver.hpp"
#include "delay_buffer.hpp"
#include "gain_interpolator.hpp"
#include "objects_renderer.hpp"
#include "ptr_adapter.hpp"
#include "variable_block_size.hpp"
