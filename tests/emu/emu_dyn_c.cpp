#include "emu_dyn.h"
RTB_EMU_DYN_DISPATCH(emu_dyn_r3, RTB_EMU_DYN(11) RTB_EMU_DYN(12))
