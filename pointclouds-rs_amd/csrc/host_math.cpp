// Host entry points of the ICP solves (the same code runs on the device inside the ICP loop).
#include "solve_math.h"

#include "pcrs_internal.h"

void pcrs_host_compose(const float ra[9], const float ta[3], const float rb[9], const float tb[3],
                       float ro[9], float to[3]) {
    pcrs_math::compose(ra, ta, rb, tb, ro, to);
}

void pcrs_host_solve_plane(const double *sums, float rot[9], float t[3]) {
    pcrs_math::Work w;
    pcrs_math::solve_plane(w, sums, rot, t);
}

void pcrs_host_solve_point(const double *sums, float rot[9], float t[3]) {
    pcrs_math::Work w;
    pcrs_math::solve_point(w, sums, rot, t);
}
