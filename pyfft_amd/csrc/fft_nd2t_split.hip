// The tiled fixed-shape N-D kernel (fft_nd2t.hpp) on split-complex parents (re / im planes): the same tile shapes as fft_nd2t.hip,
// a second translation unit so that the two build in parallel.
#define MIFFT_ND2T_SPLIT true
#define MIFFT_ND2T_NAME mifft_nd2t_split
#include "fft_nd2t.hip"
