#include "emu_dyn.h"
RTB_EMU_DYN_DISPATCH(emu_dyn_r4, RTB_EMU_DYN(13) RTB_EMU_DYN(14))
