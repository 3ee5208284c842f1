// The tiled fixed-shape N-D kernel (fft_nd2t.hpp) reading split-complex planes and writing interleaved data: the plane pass of a
// split-complex multi-pass plan (its temp buffer is interleaved).  Same tile shapes as fft_nd2t.hip.
#define MIFFT_ND2T_SPLIT true
#define MIFFT_ND2T_SPLIT_OUT false
#define MIFFT_ND2T_NAME mifft_nd2t_split_in
#include "fft_nd2t.hip"
