/* fluca_cgns.h -- field dump / checkpoint in the CGNS layout of the reference.
 *
 * What the reference writes through libcgns (cgp_*), this module writes as the same CGNS tree directly in the CGNS/HDF5
 * ("ADFH") storage layout with the HDF5 C library -- libcgns does not exist in the image, libhdf5 does.  UNVERIFIED
 * AGAINST libcgns (absent here): the node tree below follows the reference's call sequence and the CGNS SIDS file
 * mapping from the published standard; the tests pin the tree (names, labels, data types, array shapes, Fortran index
 * order) and the read-back of every number, not libcgns' acceptance of the file.
 *
 *   /Base                      CGNSBase_t  I4 [3,3]                                  cartcgns.c:18
 *     /Zone                    Zone_t      I8 (3x3): vertices N+1, cells N, 0        cartcgns.c:21-29
 *       /ZoneType              "Structured"
 *       /GridCoordinates/CoordinateX|Y|Z     R8 (N0+1, N1+1, N2+1)                  cartcgns.c:31-91
 *       /CellInfo              FlowSolution_t, GridLocation CellCenter, field Rank I4  cartcgns.c:94-116
 *       /FlowSolution<step>    FlowSolution_t, GridLocation CellCenter              cartcgns.c:355-379
 *           VelocityX|Y|Z, Pressure, PressureHalfStep      R8 (N0, N1, N2)           cartcgns.c:210-244, cnlinear.c:54,146-153
 *           /IFaceCenteredSolution  UserDefinedData_t, GridLocation IFaceCenter, FaceNormalVelocity R8 (N0+1, N1, N2)
 *           /JFaceCenteredSolution  ... JFaceCenter (N0, N1+1, N2);   /KFaceCenteredSolution ... KFaceCenter (N0, N1, N2+1)
 *                                                                                    cartcgns.c:246-291
 *       /ZoneIterativeData     FlowSolutionPointers, FlowSolutionCellInfoPointers C1 (32, nsteps)   flucacgns.c:46-56
 *     /TimeIterValues          BaseIterativeData_t I4 [nsteps], TimeValues R8 [nsteps]              flucacgns.c:41-44
 *     /SimulationType          "TimeAccurate"                                                        flucacgns.c:59
 *
 * Several ranks: the reference's ranks write hyperslabs of one file through MPI-IO; here the ranks of the NS object take
 * turns (rank 0 creates the tree, then every rank opens the file, writes its block and closes it, with NSBarrier between
 * turns) -- the same file on a file system all ranks of the node share.
 */
#ifndef FLUCA_CGNS_H
#define FLUCA_CGNS_H
#include "fluca_host.h"
#ifdef __cplusplus
extern "C" {
#endif

/* a FlucaViewer (fluca_host.h) of type FLUCAVIEWERCGNS = the reference's PETSCVIEWERFLUCACGNS: NSViewSolution, NSLoadSolution,
 * MeshView and MeshLoad (fluca_host.h) take it as they take any viewer, and dispatch through its ops table */
typedef FlucaViewer FlucaViewerCGNS;

/* PetscViewerFlucaCGNSOpen (flucacgns.c:281-315): `filename` may hold one %d, then every batch of output steps goes to
 * its own file numbered by the first step in it (-viewer_cgns_batch_size, default 1).  mode: 'w' or 'r'. */
FlErrorCode FlucaViewerCGNSOpen(const char *filename, char mode, FlucaViewerCGNS *viewer);
FlErrorCode FlucaViewerCGNSSetBatchSize(FlucaViewerCGNS viewer, int batch_size);
FlErrorCode FlucaViewerCGNSGetBatchSize(FlucaViewerCGNS viewer, int *batch_size);
/* name of the file the viewer is writing (or last wrote); borrowed */
FlErrorCode FlucaViewerCGNSGetFileName(FlucaViewerCGNS viewer, const char **filename);
/* writes TimeIterValues / ZoneIterativeData / SimulationType of the open file (rank 0) and frees the viewer (= FlucaViewerDestroy) */
FlErrorCode FlucaViewerCGNSDestroy(FlucaViewerCGNS *viewer);

/* NSViewSolution / NSLoadSolution (nssol.c:130-203) are declared in fluca_host.h: with this viewer the field links Velocity,
 * FaceNormalVelocity, Pressure and the type's PressureHalfStep become FlowSolution<step> of the current step and time (the mesh is
 * written first if the file is new; the ranks take turns), and the LAST FlowSolution of a file is read back with its step and time. */
/* NSMonitorSolution (nsmon.c:91-100) as an NSMonitorSet callback: ctx = a FlucaCGNSMonitor */
typedef struct {
  FlucaViewerCGNS viewer;
  int             view_interval; /* -ns_monitor_solution_interval, default 1 */
} FlucaCGNSMonitor;
FlErrorCode NSMonitorSolutionCGNS(NS ns, void *ctx);

/* ---- the writer / reader underneath, on HOST arrays (what the tests drive without a GPU) -------------------------- */
typedef struct {
  int64_t N[3];        /* global cells */
  int     periodic[3];
  int     rank, size;
  int     first[3], last[3]; /* this rank is the first / last along the axis */
  int64_t lo[3], len[3];     /* owned cells */
} FlucaCGNSLayout;
/* rank 0: new file with Base, Zone, coordinates (all of them, from the global face coordinates) and an empty CellInfo/Rank */
FlErrorCode FlucaCGNSCreateFile(const char *filename, const FlucaCGNSLayout *lay, const double *xf, const double *yf, const double *zf);
/* every rank: its block of CellInfo/Rank */
FlErrorCode FlucaCGNSWriteCellInfo(const char *filename, const FlucaCGNSLayout *lay);
/* rank 0: FlowSolution<step> with empty arrays for the named cell fields and face fields */
FlErrorCode FlucaCGNSCreateSolution(const char *filename, const FlucaCGNSLayout *lay, int64_t step, int ncell, const char *const cellnames[], int nface, const char *const facenames[]);
/* every rank: block of a cell field ((k*len1 + j)*len0 + i) / of the three face arrays of a face field (DMStag ownership:
 * one extra face on the last rank of a non-periodic axis; on a periodic axis the file's face N repeats face 0) */
FlErrorCode FlucaCGNSWriteCellField(const char *filename, const FlucaCGNSLayout *lay, int64_t step, const char *name, const double *data);
FlErrorCode FlucaCGNSWriteFaceField(const char *filename, const FlucaCGNSLayout *lay, int64_t step, const char *name, const double *const data[3]);
/* rank 0, when the file is complete */
FlErrorCode FlucaCGNSWriteIterativeData(const char *filename, int nsteps, const int64_t steps[], const double times[]);
/* reading */
FlErrorCode FlucaCGNSReadInfo(const char *filename, int64_t N[3], int64_t *last_step, double *last_time, int *nsteps);
FlErrorCode FlucaCGNSReadCoordinates(const char *filename, double *xf, double *yf, double *zf);
FlErrorCode FlucaCGNSReadCellField(const char *filename, const FlucaCGNSLayout *lay, int64_t step, const char *name, double *data);
FlErrorCode FlucaCGNSReadFaceField(const char *filename, const FlucaCGNSLayout *lay, int64_t step, const char *name, double *const data[3]);

#ifdef __cplusplus
}
#endif
#endif
