#include "emu_ik_wave.h"
RTB_EMU_IK_DISPATCH(emu_ik_wave_lo, RTB_EMU_IK(1) RTB_EMU_IK(2) RTB_EMU_IK(3) RTB_EMU_IK(4) RTB_EMU_IK(5) RTB_EMU_IK(6) RTB_EMU_IK(7))
