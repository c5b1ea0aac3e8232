// STUB (tests/quda_stub/README.md): the MPI names include/mugiq_hip_quda_adapter.hpp uses; declarations only.
#pragma once
typedef int MPI_Comm;
typedef int MPI_Datatype;
typedef int MPI_Op;
struct MPI_Status;
#define MPI_COMM_WORLD 1
#define MPI_COMM_NULL 0
#define MPI_DOUBLE 2
#define MPI_FLOAT 3
#define MPI_BYTE 4
#define MPI_SUM 5
#define MPI_SUCCESS 0
#define MPI_STATUS_IGNORE ((MPI_Status *)0)
int MPI_Comm_split(MPI_Comm, int, int, MPI_Comm *);
int MPI_Comm_free(MPI_Comm *);
int MPI_Sendrecv(const void *, int, MPI_Datatype, int, int, void *, int, MPI_Datatype, int, int, MPI_Comm, MPI_Status *);
int MPI_Reduce(const void *, void *, int, MPI_Datatype, MPI_Op, int, MPI_Comm);
int MPI_Gather(const void *, int, MPI_Datatype, void *, int, MPI_Datatype, int, MPI_Comm);
int MPI_Bcast(void *, int, MPI_Datatype, int, MPI_Comm);
