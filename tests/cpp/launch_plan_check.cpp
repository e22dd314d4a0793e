// Host replay of bifurcationkit.jl_amd/csrc/launch_plan.h: reads "nz tiles resident split" lines, prints the chosen z-chunk.
#include <cstdio>

#include "../../bifurcationkit.jl_amd/csrc/launch_plan.h"

int main() {
    int nz, tiles, split;
    long resident;
    while (std::scanf("%d %d %ld %d", &nz, &tiles, &resident, &split) == 4)
        std::printf("%d\n", bk::sh_plan_zchunk(nz, tiles, resident, split != 0));
    return 0;
}
