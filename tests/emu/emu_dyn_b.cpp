#include "emu_dyn.h"
RTB_EMU_DYN_DISPATCH(emu_dyn_r2, RTB_EMU_DYN(8) RTB_EMU_DYN(9) RTB_EMU_DYN(10))
