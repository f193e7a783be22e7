/* host stand-in for the CUDA toolkit header of the same name (see mfref_cuda.h); test infrastructure only */
#include "mfref_cuda.h"
