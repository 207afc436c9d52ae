// Operator codes of clMathOp / clMathConst (values of the reference's include/clenabled/clMathOpTypes.h:11-20; they travel
// through GRC make strings as plain integers, so only the values and names matter).
#pragma once
constexpr int MATHOP_MULTIPLY = 1, MATHOP_ADD = 2, MATHOP_SUBTRACT = 3, MATHOP_COMPLEX_CONJUGATE = 4, MATHOP_MULTIPLY_CONJUGATE = 5, MATHOP_LOG10 = 6, MATHOP_LOG = 7, MATHOP_SNR_HELPER = 8, MATHOP_EMPTY_W_COPY = 254, MATHOP_EMPTY = 255;
