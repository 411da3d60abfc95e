// How many CPUs this process may really use: the scheduler affinity mask cut by the cgroup CPU quota (v2 cpu.max, v1
// cfs_quota_us / cfs_period_us).  std::thread::hardware_concurrency() reports the host's logical CPUs -- 256 on the
// MI355X boxes -- while a container's lease may be 16 (profiles/r04_cpu_baseline_box.json): 64 parser threads on a
// 16-CPU quota are throttled by CFS in bursts, which is where the 2x run-to-run spread of the loader came from.
#pragma once
#include <sched.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <thread>

static inline int m6a_usable_cpus()
{
    static const int cached = [] {
        int n = (int)std::thread::hardware_concurrency();
        cpu_set_t set;
        if (sched_getaffinity(0, sizeof(set), &set) == 0) n = CPU_COUNT(&set);
        double quota = 0;
        if (FILE *f = fopen("/sys/fs/cgroup/cpu.max", "r")) {
            char q[32] = {0};
            double per = 0;
            if (fscanf(f, "%31s %lf", q, &per) == 2 && q[0] != 'm' && per > 0) quota = atof(q) / per;
            fclose(f);
        } else {
            double q = -1, per = 0;
            if (FILE *a = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) { if (fscanf(a, "%lf", &q) != 1) q = -1; fclose(a); }
            if (FILE *b = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) { if (fscanf(b, "%lf", &per) != 1) per = 0; fclose(b); }
            if (q > 0 && per > 0) quota = q / per;
        }
        if (quota > 0) n = std::min(n, (int)(quota + 0.999));
        return std::max(1, n);
    }();
    return cached;
}
