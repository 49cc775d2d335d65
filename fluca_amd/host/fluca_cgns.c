/* fluca_cgns.c -- see include/fluca_cgns.h.  The CGNS tree of the reference's viewer (fluca/src/viewer/impl/flucacgns,
 * fluca/src/mesh/impl/cart/cartcgns.c) written in the CGNS/HDF5 storage layout with libhdf5.
 *
 * CGNS/HDF5 storage of one node: an HDF5 group called by the node's name with the string attributes "name", "label"
 * (33 bytes) and "type" (3 bytes: MT, I4, I8, R4, R8, C1), an int32 attribute "flags", and -- unless the type is MT -- a
 * dataset " data" holding the node's array with the dimensions REVERSED (CGNS arrays are Fortran-ordered, first index
 * fastest).  Children are kept in creation order (link creation order tracked).  The root group carries
 * name "HDF5 MotherNode", label "Root Node of HDF5 File", type MT and the datasets " format" and " hdf5version".
 */
#include "../../include/fluca_cgns.h"
#include "../../include/fluca_host_impl.h" /* struct _p_FlucaViewer: this file implements a viewer type */

#include <hdf5.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define E_ARG_NULL 85
#define E_ARG_OUTOFRANGE 63
#define E_ARG_WRONGSTATE 73
#define E_ARG_WRONG 62
#define E_FILE_OPEN 65
#define E_FILE_WRITE 67
#define E_FILE_READ 66
#define E_FILE_UNEXPECTED 79
#define E_MEM 55
#define E_SUP 56
#define E_LIB 76
#define FLCHK(c) \
  do { \
    FlErrorCode e_ = (c); \
    if (e_) return e_; \
  } while (0)
#define CGNS_FILE_VERSION 4.2f /* CGNSLibraryVersion written into new files */

/* ------------------------------------------------------------------------------------------------ node helpers */

static int str_attr(hid_t obj, const char *key, const char *val, size_t width)
{
  char  buf[64] = {0};
  hid_t t = H5Tcopy(H5T_C_S1), s = H5Screate(H5S_SCALAR), a;
  snprintf(buf, sizeof(buf), "%s", val);
  H5Tset_size(t, width);
  a = H5Acreate2(obj, key, t, s, H5P_DEFAULT, H5P_DEFAULT);
  if (a < 0) return -1;
  const int rc = H5Awrite(a, t, buf) < 0 ? -1 : 0;
  H5Aclose(a);
  H5Sclose(s);
  H5Tclose(t);
  return rc;
}

static int node_attrs(hid_t g, const char *name, const char *label, const char *type, int with_flags)
{
  const hsize_t one = 1;
  const int32_t flags = 1;
  if (str_attr(g, "name", name, 33) || str_attr(g, "label", label, 33) || str_attr(g, "type", type, 3)) return -1;
  if (!with_flags) return 0; /* the root node carries none */
  hid_t s = H5Screate_simple(1, &one, NULL), a = H5Acreate2(g, "flags", H5T_NATIVE_INT32, s, H5P_DEFAULT, H5P_DEFAULT);
  if (a < 0) return -1;
  const int rc = H5Awrite(a, H5T_NATIVE_INT32, &flags) < 0 ? -1 : 0;
  H5Aclose(a);
  H5Sclose(s);
  return rc;
}

/* new node under parent; returns the open group (caller closes) */
static hid_t node_new(hid_t parent, const char *name, const char *label, const char *type)
{
  hid_t gcpl = H5Pcreate(H5P_GROUP_CREATE);
  H5Pset_link_creation_order(gcpl, H5P_CRT_ORDER_TRACKED | H5P_CRT_ORDER_INDEXED);
  hid_t g = H5Gcreate2(parent, name, H5P_DEFAULT, gcpl, H5P_DEFAULT);
  H5Pclose(gcpl);
  if (g < 0) return -1;
  if (node_attrs(g, name, label, type, 1)) {
    H5Gclose(g);
    return -1;
  }
  return g;
}

static hid_t h5type(const char *type)
{
  if (!strcmp(type, "I4")) return H5T_NATIVE_INT32;
  if (!strcmp(type, "I8")) return H5T_NATIVE_INT64;
  if (!strcmp(type, "R4")) return H5T_NATIVE_FLOAT;
  if (!strcmp(type, "R8")) return H5T_NATIVE_DOUBLE;
  if (!strcmp(type, "C1")) return H5T_NATIVE_INT8;
  return -1;
}

/* " data" of a node: ndim CGNS (Fortran-order) dimensions; data may be NULL (allocated, written later by blocks) */
static int node_data(hid_t g, const char *type, int ndim, const int64_t dims[], const void *data)
{
  hsize_t hd[4];
  for (int d = 0; d < ndim; ++d) hd[d] = (hsize_t)dims[ndim - 1 - d];
  hid_t s = H5Screate_simple(ndim, hd, NULL), t = h5type(type);
  hid_t ds = H5Dcreate2(g, " data", t, s, H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT);
  int   rc = ds < 0 ? -1 : 0;
  if (!rc && data) rc = H5Dwrite(ds, t, H5S_ALL, H5S_ALL, H5P_DEFAULT, data) < 0 ? -1 : 0;
  if (ds >= 0) H5Dclose(ds);
  H5Sclose(s);
  return rc;
}

static int node_with_data(hid_t parent, const char *name, const char *label, const char *type, int ndim, const int64_t dims[], const void *data)
{
  hid_t g = node_new(parent, name, label, type);
  if (g < 0) return -1;
  const int rc = ndim > 0 ? node_data(g, type, ndim, dims, data) : 0;
  H5Gclose(g);
  return rc;
}

static int node_string(hid_t parent, const char *name, const char *label, const char *value)
{
  const int64_t n = (int64_t)strlen(value);
  return node_with_data(parent, name, label, "C1", 1, &n, value);
}

/* block of an existing node's " data": CGNS-order offset/count (ndim 3), memory = contiguous block of mdims (CGNS order)
 * of which the sub-block moff/count is transferred */
static int node_block_io(hid_t file, const char *path, hid_t memtype, int write, const int64_t off[3], const int64_t count[3], const int64_t mdims[3], const int64_t moff[3], void *mem)
{
  char dpath[512];
  snprintf(dpath, sizeof(dpath), "%s/ data", path);
  hid_t ds = H5Dopen2(file, dpath, H5P_DEFAULT);
  if (ds < 0) return -1;
  hsize_t fo[3], fc[3], md[3], mo[3];
  for (int d = 0; d < 3; ++d) {
    fo[d] = (hsize_t)off[2 - d];
    fc[d] = (hsize_t)count[2 - d];
    md[d] = (hsize_t)mdims[2 - d];
    mo[d] = (hsize_t)moff[2 - d];
  }
  int rc = 0;
  if (fc[0] * fc[1] * fc[2] > 0) {
    hid_t fs = H5Dget_space(ds), ms = H5Screate_simple(3, md, NULL);
    hsize_t cur[3];
    if (H5Sget_simple_extent_ndims(fs) != 3 || H5Sget_simple_extent_dims(fs, cur, NULL) < 0) rc = -1;
    for (int d = 0; d < 3 && !rc; ++d)
      if (fo[d] + fc[d] > cur[d]) rc = -1;
    if (!rc && (H5Sselect_hyperslab(fs, H5S_SELECT_SET, fo, NULL, fc, NULL) < 0 || H5Sselect_hyperslab(ms, H5S_SELECT_SET, mo, NULL, fc, NULL) < 0)) rc = -1;
    if (!rc) rc = (write ? H5Dwrite(ds, memtype, ms, fs, H5P_DEFAULT, mem) : H5Dread(ds, memtype, ms, fs, H5P_DEFAULT, mem)) < 0 ? -1 : 0;
    H5Sclose(ms);
    H5Sclose(fs);
  }
  H5Dclose(ds);
  return rc;
}

static int read_label(hid_t file, const char *path, char out[33])
{
  hid_t g = H5Gopen2(file, path, H5P_DEFAULT);
  if (g < 0) return -1;
  hid_t a = H5Aopen(g, "label", H5P_DEFAULT);
  int   rc = -1;
  if (a >= 0) {
    hid_t t = H5Tcopy(H5T_C_S1);
    H5Tset_size(t, 33);
    memset(out, 0, 33);
    rc = H5Aread(a, t, out) < 0 ? -1 : 0;
    H5Tclose(t);
    H5Aclose(a);
  }
  H5Gclose(g);
  return rc;
}

static void sol_name(char out[40], int64_t step) { snprintf(out, 40, "FlowSolution%lld", (long long)step); }

static const char *const face_sol_names[3] = {"IFaceCenteredSolution", "JFaceCenteredSolution", "KFaceCenteredSolution"}; /* cartcgns.c:5 */
static const char *const face_sol_locs[3]  = {"IFaceCenter", "JFaceCenter", "KFaceCenter"};                               /* cartcgns.c:6 */

/* The ranks take turns on the file and close it between turns: HDF5's advisory file lock adds nothing, and flock() is not
 * available on every shared file system.  An explicit setting of the user's wins. */
static void no_hdf5_file_locking(void) { setenv("HDF5_USE_FILE_LOCKING", "FALSE", 0); }

static int layout_ok(const FlucaCGNSLayout *l)
{
  if (!l) return 0;
  for (int d = 0; d < 3; ++d)
    if (l->N[d] < 1 || l->lo[d] < 0 || l->len[d] < 0 || l->lo[d] + l->len[d] > l->N[d]) return 0;
  return l->size >= 1 && l->rank >= 0 && l->rank < l->size;
}

/* ------------------------------------------------------------------------------------------------ writer */

FlErrorCode FlucaCGNSCreateFile(const char *filename, const FlucaCGNSLayout *lay, const double *xf, const double *yf, const double *zf)
{
  if (!filename || !xf || !yf || !zf) return E_ARG_NULL;
  if (!layout_ok(lay)) return E_ARG_OUTOFRANGE;
  no_hdf5_file_locking();
  hid_t fcpl = H5Pcreate(H5P_FILE_CREATE);
  H5Pset_link_creation_order(fcpl, H5P_CRT_ORDER_TRACKED | H5P_CRT_ORDER_INDEXED);
  hid_t f = H5Fcreate(filename, H5F_ACC_TRUNC, fcpl, H5P_DEFAULT);
  H5Pclose(fcpl);
  if (f < 0) return E_FILE_OPEN;
  FlErrorCode rc = E_FILE_WRITE;
  hid_t       root = H5Gopen2(f, "/", H5P_DEFAULT), base = -1, zone = -1, gc = -1, ci = -1;
  double     *buf = NULL;
  do {
    /* root node */
    if (node_attrs(root, "HDF5 MotherNode", "Root Node of HDF5 File", "MT", 0)) break;
    {
      const char    fmt[] = "IEEE_LITTLE_32";
      char          ver[33] = {0};
      unsigned      maj, min, rel;
      const hsize_t nf = sizeof(fmt), nv = sizeof(ver);
      H5get_libversion(&maj, &min, &rel);
      snprintf(ver, sizeof(ver), "HDF5 Version %u.%u.%u", maj, min, rel);
      hid_t s = H5Screate_simple(1, &nf, NULL), ds = H5Dcreate2(root, " format", H5T_NATIVE_INT8, s, H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT);
      if (ds < 0 || H5Dwrite(ds, H5T_NATIVE_INT8, H5S_ALL, H5S_ALL, H5P_DEFAULT, fmt) < 0) break;
      H5Dclose(ds);
      H5Sclose(s);
      s  = H5Screate_simple(1, &nv, NULL);
      ds = H5Dcreate2(root, " hdf5version", H5T_NATIVE_INT8, s, H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT);
      if (ds < 0 || H5Dwrite(ds, H5T_NATIVE_INT8, H5S_ALL, H5S_ALL, H5P_DEFAULT, ver) < 0) break;
      H5Dclose(ds);
      H5Sclose(s);
    }
    {
      const float   v = CGNS_FILE_VERSION;
      const int64_t one = 1;
      if (node_with_data(root, "CGNSLibraryVersion", "CGNSLibraryVersion_t", "R4", 1, &one, &v)) break;
    }
    /* cg_base_write(..., "Base", dim, dim), cartcgns.c:18 */
    {
      const int32_t dims[2] = {3, 3};
      const int64_t two = 2;
      base = node_new(root, "Base", "CGNSBase_t", "I4");
      if (base < 0 || node_data(base, "I4", 1, &two, dims)) break;
    }
    /* cg_zone_write(..., "Zone", size, Structured), cartcgns.c:21-29: size = vertices, cells, boundary vertices (0) */
    {
      int64_t       size[9] = {0};
      const int64_t zd[2] = {3, 3};
      for (int d = 0; d < 3; ++d) {
        size[d]     = lay->N[d] + 1;
        size[3 + d] = lay->N[d];
      }
      zone = node_new(base, "Zone", "Zone_t", "I8");
      if (zone < 0 || node_data(zone, "I8", 2, zd, size)) break;
      if (node_string(zone, "ZoneType", "ZoneType_t", "Structured")) break;
    }
    /* coordinates on the vertices, cartcgns.c:31-91: e[d] = face coordinate of axis d at vertex (i0,i1,i2) */
    {
      const int64_t nv[3] = {lay->N[0] + 1, lay->N[1] + 1, lay->N[2] + 1};
      const double *xfs[3] = {xf, yf, zf};
      const char   *names[3] = {"CoordinateX", "CoordinateY", "CoordinateZ"};
      int           bad = 0;
      gc = node_new(zone, "GridCoordinates", "GridCoordinates_t", "MT");
      buf = (double *)malloc(sizeof(double) * (size_t)(nv[0] * nv[1] * nv[2]));
      if (gc < 0 || !buf) break;
      for (int d = 0; d < 3 && !bad; ++d) {
        int64_t i[3], cnt = 0;
        for (i[2] = 0; i[2] < nv[2]; ++i[2])
          for (i[1] = 0; i[1] < nv[1]; ++i[1])
            for (i[0] = 0; i[0] < nv[0]; ++i[0]) buf[cnt++] = xfs[d][i[d]];
        bad = node_with_data(gc, names[d], "DataArray_t", "R8", 3, nv, buf);
      }
      if (bad) break;
    }
    /* cell info, cartcgns.c:94-116 */
    ci = node_new(zone, "CellInfo", "FlowSolution_t", "MT");
    if (ci < 0 || node_string(ci, "GridLocation", "GridLocation_t", "CellCenter")) break;
    if (node_with_data(ci, "Rank", "DataArray_t", "I4", 3, lay->N, NULL)) break;
    rc = 0;
  } while (0);
  free(buf);
  if (ci >= 0) H5Gclose(ci);
  if (gc >= 0) H5Gclose(gc);
  if (zone >= 0) H5Gclose(zone);
  if (base >= 0) H5Gclose(base);
  H5Gclose(root);
  if (H5Fclose(f) < 0 && !rc) rc = E_FILE_WRITE;
  return rc;
}

FlErrorCode FlucaCGNSWriteCellInfo(const char *filename, const FlucaCGNSLayout *lay)
{
  if (!filename) return E_ARG_NULL;
  if (!layout_ok(lay)) return E_ARG_OUTOFRANGE;
  const int64_t n = lay->len[0] * lay->len[1] * lay->len[2], zero[3] = {0, 0, 0};
  int32_t      *e = (int32_t *)malloc(sizeof(int32_t) * (size_t)(n > 0 ? n : 1));
  if (!e) return E_MEM;
  for (int64_t i = 0; i < n; ++i) e[i] = lay->rank;
  hid_t f = H5Fopen(filename, H5F_ACC_RDWR, H5P_DEFAULT);
  if (f < 0) {
    free(e);
    return E_FILE_OPEN;
  }
  int rc = node_block_io(f, "/Base/Zone/CellInfo/Rank", H5T_NATIVE_INT32, 1, lay->lo, lay->len, lay->len, zero, e);
  free(e);
  if (H5Fclose(f) < 0) rc = -1;
  return rc ? E_FILE_WRITE : 0;
}

FlErrorCode FlucaCGNSCreateSolution(const char *filename, const FlucaCGNSLayout *lay, int64_t step, int ncell, const char *const cellnames[], int nface, const char *const facenames[])
{
  if (!filename || (ncell > 0 && !cellnames) || (nface > 0 && !facenames)) return E_ARG_NULL;
  if (!layout_ok(lay) || step < 0) return E_ARG_OUTOFRANGE;
  hid_t f = H5Fopen(filename, H5F_ACC_RDWR, H5P_DEFAULT);
  if (f < 0) return E_FILE_OPEN;
  FlErrorCode rc = E_FILE_WRITE;
  char        name[40];
  hid_t       zone = H5Gopen2(f, "/Base/Zone", H5P_DEFAULT), sol = -1;
  sol_name(name, step);
  do {
    if (zone < 0) break;
    /* cg_sol_write(sol_name, CellCenter) then the three user-data nodes in I, J, K order, cartcgns.c:355-379 */
    sol = node_new(zone, name, "FlowSolution_t", "MT");
    if (sol < 0 || node_string(sol, "GridLocation", "GridLocation_t", "CellCenter")) break;
    int bad = 0;
    for (int l = 0; l < 3 && !bad; ++l) {
      hid_t u = node_new(sol, face_sol_names[l], "UserDefinedData_t", "MT");
      bad = u < 0 || node_string(u, "GridLocation", "GridLocation_t", face_sol_locs[l]);
      for (int q = 0; q < nface && !bad; ++q) {
        int64_t dims[3] = {lay->N[0], lay->N[1], lay->N[2]};
        dims[l] += 1; /* array_size[d] = M[d] + (d == l), cartcgns.c:266 */
        bad = node_with_data(u, facenames[q], "DataArray_t", "R8", 3, dims, NULL);
      }
      if (u >= 0) H5Gclose(u);
    }
    for (int q = 0; q < ncell && !bad; ++q) bad = node_with_data(sol, cellnames[q], "DataArray_t", "R8", 3, lay->N, NULL);
    if (bad) break;
    rc = 0;
  } while (0);
  if (sol >= 0) H5Gclose(sol);
  if (zone >= 0) H5Gclose(zone);
  if (H5Fclose(f) < 0 && !rc) rc = E_FILE_WRITE;
  return rc;
}

static FlErrorCode cell_field_io(const char *filename, const FlucaCGNSLayout *lay, int64_t step, const char *name, double *data, int write)
{
  if (!filename || !name || !data) return E_ARG_NULL;
  if (!layout_ok(lay)) return E_ARG_OUTOFRANGE;
  char          sn[40], path[256];
  const int64_t zero[3] = {0, 0, 0};
  sol_name(sn, step);
  snprintf(path, sizeof(path), "/Base/Zone/%s/%s", sn, name);
  hid_t f = H5Fopen(filename, write ? H5F_ACC_RDWR : H5F_ACC_RDONLY, H5P_DEFAULT);
  if (f < 0) return E_FILE_OPEN;
  int rc = node_block_io(f, path, H5T_NATIVE_DOUBLE, write, lay->lo, lay->len, lay->len, zero, data);
  if (H5Fclose(f) < 0) rc = -1;
  return rc ? (write ? E_FILE_WRITE : E_FILE_READ) : 0;
}

static FlErrorCode face_field_io(const char *filename, const FlucaCGNSLayout *lay, int64_t step, const char *name, double *const data[3], int write)
{
  if (!filename || !name || !data) return E_ARG_NULL;
  if (!layout_ok(lay)) return E_ARG_OUTOFRANGE;
  char          sn[40], path[256];
  const int64_t zero[3] = {0, 0, 0};
  sol_name(sn, step);
  hid_t f = H5Fopen(filename, write ? H5F_ACC_RDWR : H5F_ACC_RDONLY, H5P_DEFAULT);
  if (f < 0) return E_FILE_OPEN;
  int rc = 0;
  for (int l = 0; l < 3 && !rc; ++l) {
    if (!data[l]) {
      rc = -1;
      break;
    }
    /* owned faces along l: len + 1 on the last rank of a non-periodic axis (DMStag), cartcgns.c:268-270 */
    int64_t cnt[3] = {lay->len[0], lay->len[1], lay->len[2]};
    cnt[l] += (lay->last[l] && !lay->periodic[l]) ? 1 : 0;
    snprintf(path, sizeof(path), "/Base/Zone/%s/%s/%s", sn, face_sol_names[l], name);
    rc = node_block_io(f, path, H5T_NATIVE_DOUBLE, write, lay->lo, cnt, cnt, zero, data[l]);
    /* periodic axis: the file holds N+1 faces, the last one is face 0 again (the reference reads it from the ghost layer of
     * the last rank); the rank that owns face 0 writes it */
    if (!rc && write && lay->periodic[l] && lay->first[l]) {
      int64_t off[3] = {lay->lo[0], lay->lo[1], lay->lo[2]}, one[3] = {cnt[0], cnt[1], cnt[2]};
      off[l] = lay->N[l];
      one[l] = 1;
      rc     = node_block_io(f, path, H5T_NATIVE_DOUBLE, 1, off, one, cnt, zero, data[l]);
    }
  }
  if (H5Fclose(f) < 0) rc = -1;
  return rc ? (write ? E_FILE_WRITE : E_FILE_READ) : 0;
}

FlErrorCode FlucaCGNSWriteCellField(const char *filename, const FlucaCGNSLayout *lay, int64_t step, const char *name, const double *data) { return cell_field_io(filename, lay, step, name, (double *)data, 1); }
FlErrorCode FlucaCGNSReadCellField(const char *filename, const FlucaCGNSLayout *lay, int64_t step, const char *name, double *data) { return cell_field_io(filename, lay, step, name, data, 0); }
FlErrorCode FlucaCGNSWriteFaceField(const char *filename, const FlucaCGNSLayout *lay, int64_t step, const char *name, const double *const data[3]) { return face_field_io(filename, lay, step, name, (double *const *)data, 1); }
FlErrorCode FlucaCGNSReadFaceField(const char *filename, const FlucaCGNSLayout *lay, int64_t step, const char *name, double *const data[3]) { return face_field_io(filename, lay, step, name, data, 0); }

/* PetscViewerFileClose_FlucaCGNS_Private, flucacgns.c:22-70 */
FlErrorCode FlucaCGNSWriteIterativeData(const char *filename, int nsteps, const int64_t steps[], const double times[])
{
  if (!filename || !steps || !times) return E_ARG_NULL;
  if (nsteps < 1) return E_ARG_OUTOFRANGE;
  hid_t f = H5Fopen(filename, H5F_ACC_RDWR, H5P_DEFAULT);
  if (f < 0) return E_FILE_OPEN;
  FlErrorCode   rc = E_FILE_WRITE;
  hid_t         base = H5Gopen2(f, "/Base", H5P_DEFAULT), zone = H5Gopen2(f, "/Base/Zone", H5P_DEFAULT), bi = -1, zi = -1;
  const int     width = 32;
  char         *names = (char *)malloc((size_t)nsteps * width + 1);
  const int64_t one = 1, nt = nsteps, shape[2] = {width, nsteps};
  const int32_t ns32 = nsteps;
  do {
    if (base < 0 || zone < 0 || !names) break;
    bi = node_new(base, "TimeIterValues", "BaseIterativeData_t", "I4"); /* cg_biter_write, :41 */
    if (bi < 0 || node_data(bi, "I4", 1, &one, &ns32)) break;
    if (node_with_data(bi, "TimeValues", "DataArray_t", "R8", 1, &nt, times)) break; /* :44 */
    zi = node_new(zone, "ZoneIterativeData", "ZoneIterativeData_t", "MT");          /* cg_ziter_write, :46 */
    if (zi < 0) break;
    for (int i = 0; i < nsteps; ++i) snprintf(names + (size_t)i * width, width + 1, "FlowSolution%-20lld", (long long)steps[i]); /* :52 */
    if (node_with_data(zi, "FlowSolutionPointers", "DataArray_t", "C1", 2, shape, names)) break;
    for (int i = 0; i < nsteps; ++i) snprintf(names + (size_t)i * width, width + 1, "%-32s", "CellInfo"); /* :55 */
    if (node_with_data(zi, "FlowSolutionCellInfoPointers", "DataArray_t", "C1", 2, shape, names)) break;
    if (node_string(base, "SimulationType", "SimulationType_t", "TimeAccurate")) break; /* :59 */
    rc = 0;
  } while (0);
  free(names);
  if (zi >= 0) H5Gclose(zi);
  if (bi >= 0) H5Gclose(bi);
  if (zone >= 0) H5Gclose(zone);
  if (base >= 0) H5Gclose(base);
  if (H5Fclose(f) < 0 && !rc) rc = E_FILE_WRITE;
  return rc;
}

/* ------------------------------------------------------------------------------------------------ reader */

struct lastsol {
  hid_t   file;
  int64_t step;
  int     count;
};
static herr_t find_sol(hid_t g, const char *name, const H5L_info_t *info, void *op)
{
  struct lastsol *ls = (struct lastsol *)op;
  long long       s;
  int             used = 0;
  char            path[128], label[33];
  (void)info;
  (void)g;
  if (sscanf(name, "FlowSolution%lld%n", &s, &used) != 1 || used != (int)strlen(name)) return 0;
  snprintf(path, sizeof(path), "/Base/Zone/%s", name);
  if (read_label(ls->file, path, label) || strcmp(label, "FlowSolution_t")) return 0;
  /* "assume that the last solution is the one we want" (cartcgns.c:699): links are visited in creation order */
  ls->step = s;
  ++ls->count;
  return 0;
}

FlErrorCode FlucaCGNSReadInfo(const char *filename, int64_t N[3], int64_t *last_step, double *last_time, int *nsteps)
{
  if (!filename) return E_ARG_NULL;
  hid_t f = H5Fopen(filename, H5F_ACC_RDONLY, H5P_DEFAULT);
  if (f < 0) return E_FILE_OPEN;
  FlErrorCode rc = E_FILE_UNEXPECTED;
  do {
    char    label[33];
    int32_t bd[2];
    int64_t size[9];
    if (read_label(f, "/Base", label) || strcmp(label, "CGNSBase_t")) break;
    if (read_label(f, "/Base/Zone", label) || strcmp(label, "Zone_t")) break;
    hid_t ds = H5Dopen2(f, "/Base/ data", H5P_DEFAULT);
    if (ds < 0 || H5Dread(ds, H5T_NATIVE_INT32, H5S_ALL, H5S_ALL, H5P_DEFAULT, bd) < 0) break;
    H5Dclose(ds);
    if (bd[0] != 3) break; /* "Mesh dimension does not match CGNS cell dimension" */
    ds = H5Dopen2(f, "/Base/Zone/ data", H5P_DEFAULT);
    if (ds < 0 || H5Dread(ds, H5T_NATIVE_INT64, H5S_ALL, H5S_ALL, H5P_DEFAULT, size) < 0) break;
    H5Dclose(ds);
    if (N)
      for (int d = 0; d < 3; ++d) N[d] = size[3 + d];
    struct lastsol ls = {f, -1, 0};
    hid_t          zone = H5Gopen2(f, "/Base/Zone", H5P_DEFAULT);
    hsize_t        idx = 0;
    H5Literate(zone, H5_INDEX_CRT_ORDER, H5_ITER_INC, &idx, find_sol, &ls);
    H5Gclose(zone);
    if (last_step) *last_step = ls.step;
    int nt = 0;
    if (H5Lexists(f, "/Base/TimeIterValues", H5P_DEFAULT) > 0) {
      int32_t n32 = 0;
      ds = H5Dopen2(f, "/Base/TimeIterValues/ data", H5P_DEFAULT);
      if (ds < 0 || H5Dread(ds, H5T_NATIVE_INT32, H5S_ALL, H5S_ALL, H5P_DEFAULT, &n32) < 0) break;
      H5Dclose(ds);
      nt = n32;
      if (last_time && nt > 0) {
        double *t = (double *)malloc(sizeof(double) * (size_t)nt);
        ds = H5Dopen2(f, "/Base/TimeIterValues/TimeValues/ data", H5P_DEFAULT);
        const int bad = !t || ds < 0 || H5Dread(ds, H5T_NATIVE_DOUBLE, H5S_ALL, H5S_ALL, H5P_DEFAULT, t) < 0;
        if (!bad) *last_time = t[nt - 1]; /* sol_time = times[nsteps - 1], cartcgns.c:723 */
        if (ds >= 0) H5Dclose(ds);
        free(t);
        if (bad) break;
      }
    }
    if (nsteps) *nsteps = nt;
    rc = 0;
  } while (0);
  H5Fclose(f);
  return rc;
}

FlErrorCode FlucaCGNSReadCoordinates(const char *filename, double *xf, double *yf, double *zf)
{
  int64_t N[3];
  FLCHK(FlucaCGNSReadInfo(filename, N, NULL, NULL, NULL));
  hid_t f = H5Fopen(filename, H5F_ACC_RDONLY, H5P_DEFAULT);
  if (f < 0) return E_FILE_OPEN;
  double       *out[3] = {xf, yf, zf};
  const char   *names[3] = {"/Base/Zone/GridCoordinates/CoordinateX", "/Base/Zone/GridCoordinates/CoordinateY", "/Base/Zone/GridCoordinates/CoordinateZ"};
  const int64_t zero[3] = {0, 0, 0};
  int           rc = 0;
  for (int d = 0; d < 3 && !rc; ++d) {
    if (!out[d]) continue;
    int64_t cnt[3] = {1, 1, 1};
    cnt[d] = N[d] + 1; /* the line of vertices along axis d through vertex (0,0,0) */
    rc     = node_block_io(f, names[d], H5T_NATIVE_DOUBLE, 0, zero, cnt, cnt, zero, out[d]);
  }
  H5Fclose(f);
  return rc ? E_FILE_READ : 0;
}

/* ------------------------------------------------------------------------------------------------ viewer */

/* PetscViewer_FlucaCGNS (flucacgns.h): the data of a FlucaViewer of type FLUCAVIEWERCGNS */
#define CGNS_MAXPENDING 16
typedef struct {
  char    *tmpl;     /* filename or template */
  int      is_template, batch_size;
  int      rank;     /* of the NS that wrote last */
  char    *filename; /* the file being written (NULL: none open) */
  char    *lastname;
  int64_t  last_step;
  int      nsteps, cap;
  int64_t *steps;
  double  *times;
  /* one NSViewSolution / NSLoadSolution in flight (solutionbegin .. solutionend) */
  FlucaCGNSLayout lay;
  int64_t         step, sz[4];
  int             device, newfile, skip;
  int             ncell, nface;
  char            cellname[CGNS_MAXPENDING][33], facename[CGNS_MAXPENDING][33];
  double         *celldev[CGNS_MAXPENDING], *facedev[CGNS_MAXPENDING][3];
} ViewerCGNS;

static FlErrorCode ViewerDestroy_CGNS(FlucaViewer viewer);
static FlErrorCode ViewerViewMesh_CGNS(FlucaViewer viewer, Mesh mesh);
static FlErrorCode ViewerLoadMesh_CGNS(FlucaViewer viewer, int64_t N[3], double *xf[3]);
static FlErrorCode ViewerSolutionBegin_CGNS(FlucaViewer viewer, NS ns, int write);
static FlErrorCode ViewerCellField_CGNS(FlucaViewer viewer, NS ns, const char *name, int ncomp, double *dev);
static FlErrorCode ViewerFaceField_CGNS(FlucaViewer viewer, NS ns, const char *name, double *const dev[3]);
static FlErrorCode ViewerSolutionEnd_CGNS(FlucaViewer viewer, NS ns);

#define CGNS_DATA(viewer)                                                                  \
  if (!(viewer)) return E_ARG_NULL;                                                        \
  if (!(viewer)->type || strcmp((viewer)->type, FLUCAVIEWERCGNS)) return E_ARG_WRONG;      \
  ViewerCGNS *v = (ViewerCGNS *)(viewer)->data

FlErrorCode FlucaViewerCGNSOpen(const char *filename, char mode, FlucaViewerCGNS *viewer)
{
  if (!filename || !viewer) return E_ARG_NULL;
  if (mode != 'w' && mode != 'r') return E_ARG_WRONG; /* "Unsupported file mode", flucacgns.c:95 */
  ViewerCGNS *v = (ViewerCGNS *)calloc(1, sizeof(*v));
  if (!v) return E_MEM;
  const FlErrorCode rc = FlucaViewerCreate(FLUCAVIEWERCGNS, mode, viewer);
  if (rc) {
    free(v);
    return rc;
  }
  v->tmpl        = strdup(filename);
  v->is_template = strstr(filename, "%d") != NULL || strstr(filename, "%0") != NULL; /* flucacgns.c:185-190: a '%' makes it a template */
  v->batch_size  = 1;                                                                /* flucacgns.c:220 */
  v->last_step   = -1;
  (*viewer)->data               = v;
  (*viewer)->ops->viewmesh      = ViewerViewMesh_CGNS;
  (*viewer)->ops->loadmesh      = ViewerLoadMesh_CGNS;
  (*viewer)->ops->solutionbegin = ViewerSolutionBegin_CGNS;
  (*viewer)->ops->cellfield     = ViewerCellField_CGNS;
  (*viewer)->ops->facefield     = ViewerFaceField_CGNS;
  (*viewer)->ops->solutionend   = ViewerSolutionEnd_CGNS;
  (*viewer)->ops->destroy       = ViewerDestroy_CGNS;
  no_hdf5_file_locking();
  H5Eset_auto2(H5E_DEFAULT, NULL, NULL); /* errors are reported through return codes */
  return 0;
}
FlErrorCode FlucaViewerCGNSSetBatchSize(FlucaViewerCGNS viewer, int batch_size)
{
  CGNS_DATA(viewer);
  if (batch_size < 1) return E_ARG_OUTOFRANGE;
  v->batch_size = batch_size;
  return 0;
}
FlErrorCode FlucaViewerCGNSGetBatchSize(FlucaViewerCGNS viewer, int *batch_size)
{
  CGNS_DATA(viewer);
  if (!batch_size) return E_ARG_NULL;
  *batch_size = v->batch_size;
  return 0;
}
FlErrorCode FlucaViewerCGNSGetFileName(FlucaViewerCGNS viewer, const char **filename)
{
  CGNS_DATA(viewer);
  if (!filename) return E_ARG_NULL;
  *filename = v->filename ? v->filename : v->lastname;
  return 0;
}

/* flucacgns.c:22-70 on rank 0; every rank forgets the file */
static FlErrorCode viewer_close_file(ViewerCGNS *v, int rank)
{
  FlErrorCode rc = 0;
  if (!v->filename) return 0;
  if (v->nsteps > 0 && rank == 0) rc = FlucaCGNSWriteIterativeData(v->filename, v->nsteps, v->steps, v->times);
  free(v->lastname);
  v->lastname = v->filename;
  v->filename = NULL;
  v->nsteps   = 0;
  return rc;
}

static FlErrorCode ViewerDestroy_CGNS(FlucaViewer viewer)
{
  ViewerCGNS *v = (ViewerCGNS *)viewer->data;
  if (!v) return 0;
  const FlErrorCode rc = viewer->mode == 'w' ? viewer_close_file(v, v->rank) : 0;
  free(v->tmpl);
  free(v->filename);
  free(v->lastname);
  free(v->steps);
  free(v->times);
  free(v);
  viewer->data = NULL;
  return rc;
}
FlErrorCode FlucaViewerCGNSDestroy(FlucaViewerCGNS *viewer) { return FlucaViewerDestroy(viewer); }

static FlErrorCode mesh_layout(Mesh mesh, FlucaCGNSLayout *lay)
{
  if (!mesh) return E_ARG_WRONGSTATE;
  FLCHK(MeshCartGetGlobalSizes(mesh, &lay->N[0], &lay->N[1], &lay->N[2]));
  FLCHK(MeshCartGetCorners(mesh, &lay->lo[0], &lay->lo[1], &lay->lo[2], &lay->len[0], &lay->len[1], &lay->len[2]));
  FLCHK(MeshCartGetIsFirstRank(mesh, &lay->first[0], &lay->first[1], &lay->first[2]));
  FLCHK(MeshCartGetIsLastRank(mesh, &lay->last[0], &lay->last[1], &lay->last[2]));
  FLCHK(MeshGetRank(mesh, &lay->rank, &lay->size));
  const Mesh_Cart *cart = (const Mesh_Cart *)mesh->data;
  for (int d = 0; d < 3; ++d) lay->periodic[d] = cart->bndTypes[d] == MESHCART_BOUNDARY_PERIODIC;
  return 0;
}

/* the name of the file a write viewer opens next: the template filled with the output sequence number (flucacgns.c:82) */
static FlErrorCode viewer_open_name(ViewerCGNS *v, int64_t seq)
{
  char name[4096];
  if (v->is_template) snprintf(name, sizeof(name), v->tmpl, (int)(seq < 0 ? 0 : seq));
  else snprintf(name, sizeof(name), "%s", v->tmpl);
  v->filename = strdup(name);
  return v->filename ? 0 : E_MEM;
}

/* MeshView_Cart_CGNS (cartcgns.c:8-118): Base, Zone, the vertex coordinates and CellInfo/Rank of a file that has none yet.
 * Outside NSViewSolution the mesh has no communicator to take turns on, so a decomposed mesh is written through NSViewSolution only. */
static FlErrorCode ViewerViewMesh_CGNS(FlucaViewer viewer, Mesh mesh)
{
  ViewerCGNS *v = (ViewerCGNS *)viewer->data;
  if (viewer->mode != 'w') return E_ARG_WRONGSTATE;
  if (v->filename) return 0; /* "if (cgv->file_num && cgv->base)": the open file has its mesh (cartcgns.c:15) */
  FlucaCGNSLayout lay;
  FLCHK(mesh_layout(mesh, &lay));
  if (lay.size > 1) return E_SUP;
  const double *xf, *yf, *zf;
  FLCHK(MeshCartGetCoordinateArraysRead(mesh, &xf, &yf, &zf));
  FLCHK(viewer_open_name(v, viewer->seqnum));
  v->rank = lay.rank;
  FLCHK(FlucaCGNSCreateFile(v->filename, &lay, xf, yf, zf));
  return FlucaCGNSWriteCellInfo(v->filename, &lay);
}

/* MeshLoad_Cart_CGNS (cartcgns.c:120-158) */
static FlErrorCode ViewerLoadMesh_CGNS(FlucaViewer viewer, int64_t N[3], double *xf[3])
{
  ViewerCGNS *v = (ViewerCGNS *)viewer->data;
  if (viewer->mode != 'r' || v->is_template) return E_ARG_WRONGSTATE;
  FLCHK(FlucaCGNSReadInfo(v->tmpl, N, NULL, NULL, NULL));
  for (int d = 0; d < 3; ++d) {
    xf[d] = (double *)malloc(sizeof(double) * (size_t)(N[d] + 1));
    if (!xf[d]) return E_MEM;
  }
  const FlErrorCode rc = FlucaCGNSReadCoordinates(v->tmpl, xf[0], xf[1], xf[2]);
  if (rc)
    for (int d = 0; d < 3; ++d) {
      free(xf[d]);
      xf[d] = NULL;
    }
  return rc;
}

static FlErrorCode ViewerSolutionBegin_CGNS(FlucaViewer viewer, NS ns, int write)
{
  ViewerCGNS *v = (ViewerCGNS *)viewer->data;
  Mesh        mesh;
  FLCHK(NSGetMesh(ns, &mesh));
  FLCHK(mesh_layout(mesh, &v->lay));
  for (int d = 0; d < 3; ++d) { /* the NS boundary conditions decide what is periodic (they agree with the mesh after NSSetUp) */
    int                 idx;
    NSBoundaryCondition bc;
    FLCHK(MeshCartGetBoundaryIndex(mesh, (MeshCartBoundaryLocation)(2 * d), &idx));
    FLCHK(NSGetBoundaryCondition(ns, idx, &bc));
    v->lay.periodic[d] = bc.type == NS_BC_PERIODIC;
  }
  FLCHK(NSGetLocalSizes(ns, v->sz));
  FLCHK(NSGetDevice(ns, &v->device));
  { /* the copies below are plain blocking copies: the solver's own stream must have drained first */
    fl_poisson *poisson;
    FLCHK(NSGetPoisson(ns, &poisson));
    if (fl_poisson_synchronize(poisson)) return E_LIB;
  }
  v->ncell = v->nface = 0;
  v->skip = v->newfile = 0;
  if (!write) {
    if (v->is_template) return E_ARG_WRONGSTATE; /* PetscViewerCheckReadable */
    int64_t N[3], step = -1;
    double  t = 0.;
    int     nsteps = 0;
    FLCHK(FlucaCGNSReadInfo(v->tmpl, N, &step, &t, &nsteps));
    for (int d = 0; d < 3; ++d)
      if (N[d] != v->lay.N[d]) return E_LIB; /* "Mesh size does not match CGNS zone size", cartcgns.c:697 */
    if (step < 0 || nsteps < 1) return E_LIB; /* no FlowSolution<n> / no BaseIterativeData */
    v->step = step;
    viewer->seqnum = step; /* what VecLoad_Cart_CGNS leaves in the mesh's output sequence (cartcgns.c:736-755) */
    viewer->seqval = t;
    return 0;
  }
  double t;
  FLCHK(NSGetTimeStep(ns, &v->step));
  FLCHK(NSGetTime(ns, &t));
  v->rank = v->lay.rank; /* the destroy routine has no NS argument: it closes the file as this rank */
  if (v->last_step == v->step && v->filename) { /* this step is in the file already (cgv->sol stays set, cartcgns.c:336) */
    v->skip = 1;
    return 0;
  }
  /* PetscViewerFlucaCGNSCheckBatch_Internal, flucacgns.c:104-115 */
  if (v->is_template && v->filename && v->nsteps >= v->batch_size) FLCHK(viewer_close_file(v, v->lay.rank));
  if (!v->filename) {
    FLCHK(viewer_open_name(v, v->step));
    v->newfile = 1;
  }
  if (v->nsteps == v->cap) {
    v->cap   = v->cap ? 2 * v->cap : 20;
    v->steps = (int64_t *)realloc(v->steps, sizeof(int64_t) * (size_t)v->cap);
    v->times = (double *)realloc(v->times, sizeof(double) * (size_t)v->cap);
    if (!v->steps || !v->times) return E_MEM;
  }
  v->steps[v->nsteps] = v->step;
  v->times[v->nsteps] = t;
  ++v->nsteps;
  v->last_step = v->step;
  return 0;
}

/* VecView_Cart_Local_CGNS / VecLoad_Cart_CGNS for a cell vector (cartcgns.c:293-401, 644-758): written when the solution ends (one
 * device-to-host copy, then the ranks take turns on the file), read at once */
static FlErrorCode ViewerCellField_CGNS(FlucaViewer viewer, NS ns, const char *name, int ncomp, double *dev)
{
  ViewerCGNS *v = (ViewerCGNS *)viewer->data;
  (void)ns;
  if (!name || !dev) return E_ARG_NULL;
  if (ncomp != 1 && ncomp != 3) return E_ARG_OUTOFRANGE;
  for (int c = 0; c < ncomp; ++c) {
    char full[33];
    if (ncomp == 1) snprintf(full, sizeof(full), "%s", name);
    else snprintf(full, sizeof(full), "%s%c", name, 'X' + c); /* "%s%c", cartcgns.c:383-386 */
    double *d = dev + (size_t)c * (size_t)v->sz[0];
    if (viewer->mode == 'w') {
      if (v->skip) continue;
      if (v->ncell >= CGNS_MAXPENDING) return E_ARG_OUTOFRANGE;
      snprintf(v->cellname[v->ncell], 33, "%s", full);
      v->celldev[v->ncell++] = d;
    } else {
      double *h = (double *)malloc(sizeof(double) * (size_t)(v->sz[0] > 0 ? v->sz[0] : 1));
      if (!h) return E_MEM;
      FlErrorCode rc = FlucaCGNSReadCellField(v->tmpl, &v->lay, v->step, full, h);
      if (!rc) rc = fl_memcpy_h2d(v->device, d, h, sizeof(double) * (size_t)v->sz[0]) ? E_LIB : 0;
      free(h);
      FLCHK(rc);
    }
  }
  return 0;
}
static FlErrorCode ViewerFaceField_CGNS(FlucaViewer viewer, NS ns, const char *name, double *const dev[3])
{
  ViewerCGNS *v = (ViewerCGNS *)viewer->data;
  (void)ns;
  if (!name || !dev) return E_ARG_NULL;
  if (viewer->mode == 'w') {
    if (v->skip) return 0;
    if (v->nface >= CGNS_MAXPENDING) return E_ARG_OUTOFRANGE;
    snprintf(v->facename[v->nface], 33, "%s", name);
    for (int l = 0; l < 3; ++l) v->facedev[v->nface][l] = dev[l];
    ++v->nface;
    return 0;
  }
  double     *hf[3] = {0};
  FlErrorCode rc = 0;
  for (int l = 0; l < 3 && !rc; ++l) {
    hf[l] = (double *)malloc(sizeof(double) * (size_t)(v->sz[1 + l] > 0 ? v->sz[1 + l] : 1));
    if (!hf[l]) rc = E_MEM;
  }
  if (!rc) rc = FlucaCGNSReadFaceField(v->tmpl, &v->lay, v->step, name, hf);
  for (int l = 0; l < 3 && !rc; ++l) rc = fl_memcpy_h2d(v->device, dev[l], hf[l], sizeof(double) * (size_t)v->sz[1 + l]) ? E_LIB : 0;
  for (int l = 0; l < 3; ++l) free(hf[l]);
  return rc;
}

static FlErrorCode ViewerSolutionEnd_CGNS(FlucaViewer viewer, NS ns)
{
  ViewerCGNS *v = (ViewerCGNS *)viewer->data;
  if (viewer->mode != 'w' || v->skip) return 0;
  const FlucaCGNSLayout *lay = &v->lay;
  /* device -> host once, then the ranks take turns on the file */
  double     *hc[CGNS_MAXPENDING] = {0}, *hf[CGNS_MAXPENDING][3] = {{0}};
  const char *cellnames[CGNS_MAXPENDING], *facenames[CGNS_MAXPENDING];
  FlErrorCode rc = 0;
  for (int q = 0; q < v->ncell && !rc; ++q) {
    cellnames[q] = v->cellname[q];
    hc[q] = (double *)malloc(sizeof(double) * (size_t)(v->sz[0] > 0 ? v->sz[0] : 1));
    rc    = !hc[q] ? E_MEM : fl_memcpy_d2h(v->device, hc[q], v->celldev[q], sizeof(double) * (size_t)v->sz[0]) ? E_LIB : 0;
  }
  for (int q = 0; q < v->nface && !rc; ++q) {
    facenames[q] = v->facename[q];
    for (int l = 0; l < 3 && !rc; ++l) {
      hf[q][l] = (double *)malloc(sizeof(double) * (size_t)(v->sz[1 + l] > 0 ? v->sz[1 + l] : 1));
      rc       = !hf[q][l] ? E_MEM : fl_memcpy_d2h(v->device, hf[q][l], v->facedev[q][l], sizeof(double) * (size_t)v->sz[1 + l]) ? E_LIB : 0;
    }
  }
  for (int turn = 0; turn < lay->size; ++turn) {
    if (turn == lay->rank && !rc) {
      if (lay->rank == 0) {
        if (v->newfile) {
          Mesh          mesh;
          const double *xf, *yf, *zf;
          rc = NSGetMesh(ns, &mesh);
          if (!rc) rc = MeshCartGetCoordinateArraysRead(mesh, &xf, &yf, &zf);
          if (!rc) rc = FlucaCGNSCreateFile(v->filename, lay, xf, yf, zf);
        }
        if (!rc) rc = FlucaCGNSCreateSolution(v->filename, lay, v->step, v->ncell, cellnames, v->nface, facenames);
      }
      if (!rc && v->newfile) rc = FlucaCGNSWriteCellInfo(v->filename, lay);
      for (int q = 0; q < v->ncell && !rc; ++q) rc = FlucaCGNSWriteCellField(v->filename, lay, v->step, cellnames[q], hc[q]);
      for (int q = 0; q < v->nface && !rc; ++q) rc = FlucaCGNSWriteFaceField(v->filename, lay, v->step, facenames[q], (const double *const *)hf[q]);
    }
    if (lay->size > 1) {
      const FlErrorCode brc = NSBarrier(ns);
      if (!rc) rc = brc;
    }
  }
  for (int q = 0; q < v->ncell; ++q) free(hc[q]);
  for (int q = 0; q < v->nface; ++q)
    for (int l = 0; l < 3; ++l) free(hf[q][l]);
  return rc;
}

FlErrorCode NSMonitorSolutionCGNS(NS ns, void *ctx) /* nsmon.c:91-100 */
{
  FlucaCGNSMonitor *m = (FlucaCGNSMonitor *)ctx;
  int64_t           step;
  if (!ns || !m) return E_ARG_NULL;
  FLCHK(NSGetTimeStep(ns, &step));
  if (m->view_interval > 0 && step % m->view_interval == 0) FLCHK(NSViewSolution(ns, m->viewer));
  return 0;
}
