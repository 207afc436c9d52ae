// Every public header of the MI355X build of gr::clenabled in one include (the block layer, the CLI and the pybind module use
// this; users of the reference keep including <clenabled/clFFT.h>, <clenabled/clFilter.h>, ... one per block, exactly as
// before: the reference installs include/clenabled/<Block>.h, and so does this build -- tests/test_host_cpp.py compiles a
// translation unit per header).
#pragma once
#include "GRCLBase.h"
#include "clMathOpTypes.h"
#include "clSComplex.h"
#include "clMathOp.h"
#include "clMathConst.h"
#include "clFFT.h"
#include "clFilter.h"
#include "clComplexFilter.h"
#include "clPolyphaseChannelizer.h"
#include "clXEngine.h"
// widened rows
#include "clLog.h"
#include "clSNR.h"
#include "clComplexToMag.h"
#include "clComplexToArg.h"
#include "clComplexToMagPhase.h"
#include "clMagPhaseToComplex.h"
#include "clQuadratureDemod.h"
#include "clxcorrelate_fft_vcf.h"
