// STUB (tests/quda_stub/README.md).  Declarations only.
#pragma once
struct Topology;
int comm_rank();
int comm_size();
int comm_dim(int);
int comm_coord(int);
int comm_dim_partitioned(int);
Topology *comm_default_topology();
int comm_rank_displaced(const Topology *, const int displacement[]);
