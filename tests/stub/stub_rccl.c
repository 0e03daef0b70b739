/* A stand-in for librccl.so, for CPU tests of comm.py's error paths (tests/test_comm_errors.py): the seven entry points
 * RcclCommunicator binds, with failures on demand.  Test infrastructure only - never loaded by the package.
 *
 *   ACG_STUB_FAIL_INIT_RANK=<r>   ncclCommInitRank returns ncclSystemError (2) on rank r
 *   ACG_STUB_FAIL_ALLREDUCE=<n>   the n-th ncclAllReduce of the process (1-based) returns ncclInvalidArgument (4)
 *   ACG_STUB_FAIL_DESTROY=1       ncclCommDestroy returns ncclInternalError (3)
 * A communicator is a heap cell holding (nranks, rank); ncclAllReduce on it is the one-rank identity (in place). */
#include <stddef.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef struct { char internal[128]; } ncclUniqueId;
typedef struct { int magic, nranks, rank; } stub_comm;
static int n_allreduce = 0;

static int env_int(const char* name, int dflt) {
  const char* v = getenv(name);
  return (v && *v) ? atoi(v) : dflt;
}

/* what a real id looks like: an 8-byte magic, then a sockaddr (AF_INET = 02 00 ...): NULs from byte 9 on */
static void fill_id(ncclUniqueId* id) {
  static const unsigned char head[16] = {0x2b, 0xad, 0xf0, 0x0d, 0xde, 0xad, 0xbe, 0xef, 0x02, 0x00, 0x9c, 0x40, 0x7f, 0x00, 0x00, 0x01};
  memset(id->internal, 0, sizeof id->internal);
  memcpy(id->internal, head, sizeof head);
  id->internal[127] = 0x5a;
}

int ncclGetVersion(int* v) { *v = 99999; return 0; }
int ncclGetUniqueId(ncclUniqueId* id) { fill_id(id); return 0; }

int ncclCommInitRank(void** comm, int nranks, ncclUniqueId id, int rank) {
  ncclUniqueId want;
  fill_id(&want);
  if (memcmp(want.internal, id.internal, sizeof want.internal) != 0) return 4; /* the id did not arrive intact */
  if (rank < 0 || rank >= nranks) return 4;
  if (env_int("ACG_STUB_FAIL_INIT_RANK", -1) == rank) return 2;
  stub_comm* c = (stub_comm*)malloc(sizeof *c);
  c->magic = 0x5ccc; c->nranks = nranks; c->rank = rank;
  *comm = c;
  return 0;
}

int ncclAllReduce(const void* send, void* recv, size_t count, int dtype, int op, void* comm, void* stream) {
  (void)count; (void)dtype; (void)op; (void)stream;
  if (!comm || ((stub_comm*)comm)->magic != 0x5ccc || send != recv) return 4;
  if (++n_allreduce == env_int("ACG_STUB_FAIL_ALLREDUCE", -1)) return 4;
  return 0;
}

int ncclCommDestroy(void* comm) {
  if (!comm || ((stub_comm*)comm)->magic != 0x5ccc) return 4;
  ((stub_comm*)comm)->magic = 0;
  free(comm);
  return env_int("ACG_STUB_FAIL_DESTROY", 0) ? 3 : 0;
}

const char* ncclGetErrorString(int rc) {
  switch (rc) {
    case 0: return "no error";
    case 2: return "unhandled system error (stub)";
    case 3: return "internal error (stub)";
    case 4: return "invalid argument (stub)";
    default: return "unknown result code (stub)";
  }
}
