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

struct _p_FlucaViewerCGNS {
  char    *tmpl;     /* filename or template */
  int      is_template, batch_size;
  char     mode;
  int      rank;     /* of the NS that wrote last */
  char    *filename; /* the file being written (NULL: none open) */
  char    *lastname;
  int64_t  last_step;
  int      nsteps, cap;
  int64_t *steps;
  double  *times;
};

FlErrorCode FlucaViewerCGNSOpen(const char *filename, char mode, FlucaViewerCGNS *viewer)
{
  if (!filename || !viewer) return E_ARG_NULL;
  if (mode != 'w' && mode != 'r') return E_ARG_WRONG; /* "Unsupported file mode", flucacgns.c:95 */
  FlucaViewerCGNS v = (FlucaViewerCGNS)calloc(1, sizeof(*v));
  if (!v) return E_MEM;
  v->tmpl        = strdup(filename);
  v->is_template = strstr(filename, "%d") != NULL || strstr(filename, "%0") != NULL; /* flucacgns.c:185-190: a '%' makes it a template */
  v->batch_size  = 1;                                                                /* flucacgns.c:220 */
  v->mode        = mode;
  v->last_step   = -1;
  no_hdf5_file_locking();
  H5Eset_auto2(H5E_DEFAULT, NULL, NULL); /* errors are reported through return codes */
  *viewer = v;
  return 0;
}
FlErrorCode FlucaViewerCGNSSetBatchSize(FlucaViewerCGNS v, int batch_size)
{
  if (!v) return E_ARG_NULL;
  if (batch_size < 1) return E_ARG_OUTOFRANGE;
  v->batch_size = batch_size;
  return 0;
}
FlErrorCode FlucaViewerCGNSGetBatchSize(FlucaViewerCGNS v, int *batch_size)
{
  if (!v || !batch_size) return E_ARG_NULL;
  *batch_size = v->batch_size;
  return 0;
}
FlErrorCode FlucaViewerCGNSGetFileName(FlucaViewerCGNS v, const char **filename)
{
  if (!v || !filename) return E_ARG_NULL;
  *filename = v->filename ? v->filename : v->lastname;
  return 0;
}

/* flucacgns.c:22-70 on rank 0; every rank forgets the file */
static FlErrorCode viewer_close_file(FlucaViewerCGNS v, int rank)
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

FlErrorCode FlucaViewerCGNSDestroy(FlucaViewerCGNS *viewer)
{
  if (!viewer || !*viewer) return 0;
  FlucaViewerCGNS   v = *viewer;
  const FlErrorCode rc = v->mode == 'w' ? viewer_close_file(v, v->rank) : 0;
  free(v->tmpl);
  free(v->filename);
  free(v->lastname);
  free(v->steps);
  free(v->times);
  free(v);
  *viewer = NULL;
  return rc;
}

static FlErrorCode ns_layout(NS ns, FlucaCGNSLayout *lay)
{
  Mesh mesh;
  FLCHK(NSGetMesh(ns, &mesh));
  if (!mesh) return E_ARG_WRONGSTATE;
  FLCHK(MeshCartGetGlobalSizes(mesh, &lay->N[0], &lay->N[1], &lay->N[2]));
  FLCHK(MeshCartGetCorners(mesh, &lay->lo[0], &lay->lo[1], &lay->lo[2], &lay->len[0], &lay->len[1], &lay->len[2]));
  FLCHK(MeshCartGetIsFirstRank(mesh, &lay->first[0], &lay->first[1], &lay->first[2]));
  FLCHK(MeshCartGetIsLastRank(mesh, &lay->last[0], &lay->last[1], &lay->last[2]));
  FLCHK(MeshGetRank(mesh, &lay->rank, &lay->size));
  for (int d = 0; d < 3; ++d) {
    int idx;
    NSBoundaryCondition bc;
    FLCHK(MeshCartGetBoundaryIndex(mesh, (MeshCartBoundaryLocation)(2 * d), &idx));
    FLCHK(NSGetBoundaryCondition(ns, idx, &bc));
    lay->periodic[d] = bc.type == NS_BC_PERIODIC;
  }
  return 0;
}

static const char *const cell_fields[] = {"VelocityX", "VelocityY", "VelocityZ", "Pressure", "PressureHalfStep"}; /* nsbasic.c:180-182 + "%s%c" cartcgns.c:383-386; cnlinear.c:54 */
static const char *const face_fields[] = {"FaceNormalVelocity"};

/* the five cell arrays and three face arrays of the solution on the device, and the owned sizes */
static FlErrorCode ns_arrays(NS ns, double *cell[5], double *face[3], int64_t sz[4])
{
  double *v, *p, *ph;
  FLCHK(NSGetSolutionArrays(ns, &v, face, &p));
  FLCHK(NSGetPressureHalfStep(ns, &ph));
  FLCHK(NSGetLocalSizes(ns, sz));
  for (int c = 0; c < 3; ++c) cell[c] = v + (size_t)c * (size_t)sz[0];
  cell[3] = p;
  cell[4] = ph;
  /* the copies below are plain blocking copies: the solver's own stream must have drained first */
  fl_poisson *poisson;
  FLCHK(NSGetPoisson(ns, &poisson));
  if (fl_poisson_synchronize(poisson)) return E_LIB;
  return 0;
}

FlErrorCode NSViewSolution(NS ns, FlucaViewerCGNS v)
{
  if (!ns || !v) return E_ARG_NULL;
  if (v->mode != 'w') return E_ARG_WRONGSTATE;
  FlucaCGNSLayout lay;
  int64_t         step, sz[4];
  double          t, *cell[5], *face[3];
  int             device;
  FLCHK(ns_layout(ns, &lay));
  FLCHK(ns_arrays(ns, cell, face, sz));
  FLCHK(NSGetTimeStep(ns, &step));
  FLCHK(NSGetTime(ns, &t));
  FLCHK(NSGetDevice(ns, &device));
  v->rank = lay.rank; /* FlucaViewerCGNSDestroy has no NS argument: it closes the file as this rank */
  if (v->last_step == step && v->filename) return 0; /* this step is in the file already (cgv->sol stays set, cartcgns.c:336) */

  /* PetscViewerFlucaCGNSCheckBatch_Internal, flucacgns.c:104-115 */
  if (v->is_template && v->filename && v->nsteps >= v->batch_size) FLCHK(viewer_close_file(v, lay.rank));
  int newfile = 0;
  if (!v->filename) {
    char name[4096];
    if (v->is_template) snprintf(name, sizeof(name), v->tmpl, (int)step); /* flucacgns.c:82 */
    else snprintf(name, sizeof(name), "%s", v->tmpl);
    v->filename = strdup(name);
    newfile     = 1;
  }
  if (v->nsteps == v->cap) {
    v->cap   = v->cap ? 2 * v->cap : 20;
    v->steps = (int64_t *)realloc(v->steps, sizeof(int64_t) * (size_t)v->cap);
    v->times = (double *)realloc(v->times, sizeof(double) * (size_t)v->cap);
    if (!v->steps || !v->times) return E_MEM;
  }
  v->steps[v->nsteps] = step;
  v->times[v->nsteps] = t;
  ++v->nsteps;
  v->last_step = step;

  /* device -> host once, then the ranks take turns on the file */
  double *hc[5] = {0}, *hf[3] = {0};
  FlErrorCode rc = 0;
  for (int q = 0; q < 5 && !rc; ++q) {
    hc[q] = (double *)malloc(sizeof(double) * (size_t)(sz[0] > 0 ? sz[0] : 1));
    rc    = !hc[q] ? E_MEM : fl_memcpy_d2h(device, hc[q], cell[q], sizeof(double) * (size_t)sz[0]);
  }
  for (int l = 0; l < 3 && !rc; ++l) {
    hf[l] = (double *)malloc(sizeof(double) * (size_t)(sz[1 + l] > 0 ? sz[1 + l] : 1));
    rc    = !hf[l] ? E_MEM : fl_memcpy_d2h(device, hf[l], face[l], sizeof(double) * (size_t)sz[1 + l]);
  }
  for (int turn = 0; turn < lay.size; ++turn) {
    if (turn == lay.rank && !rc) {
      if (lay.rank == 0) {
        if (newfile) {
          Mesh          mesh;
          const double *xf, *yf, *zf;
          rc = NSGetMesh(ns, &mesh);
          if (!rc) rc = MeshCartGetCoordinateArraysRead(mesh, &xf, &yf, &zf);
          if (!rc) rc = FlucaCGNSCreateFile(v->filename, &lay, xf, yf, zf);
        }
        if (!rc) rc = FlucaCGNSCreateSolution(v->filename, &lay, step, 5, cell_fields, 1, face_fields);
      }
      if (!rc && newfile) rc = FlucaCGNSWriteCellInfo(v->filename, &lay);
      /* field order of NSViewSolution: Velocity, FaceNormalVelocity, Pressure (field links), then PressureHalfStep */
      for (int q = 0; q < 3 && !rc; ++q) rc = FlucaCGNSWriteCellField(v->filename, &lay, step, cell_fields[q], hc[q]);
      if (!rc) rc = FlucaCGNSWriteFaceField(v->filename, &lay, step, face_fields[0], (const double *const *)hf);
      for (int q = 3; q < 5 && !rc; ++q) rc = FlucaCGNSWriteCellField(v->filename, &lay, step, cell_fields[q], hc[q]);
    }
    if (lay.size > 1) {
      const FlErrorCode brc = NSBarrier(ns);
      if (!rc) rc = brc;
    }
  }
  for (int q = 0; q < 5; ++q) free(hc[q]);
  for (int l = 0; l < 3; ++l) free(hf[l]);
  return rc;
}

FlErrorCode NSLoadSolution(NS ns, FlucaViewerCGNS v)
{
  if (!ns || !v) return E_ARG_NULL;
  if (v->mode != 'r' || v->is_template) return E_ARG_WRONGSTATE; /* PetscViewerCheckReadable */
  FlucaCGNSLayout lay;
  int64_t         N[3], step = -1, sz[4];
  double          t = 0., *cell[5], *face[3];
  int             device, nsteps = 0;
  FLCHK(ns_layout(ns, &lay));
  FLCHK(ns_arrays(ns, cell, face, sz));
  FLCHK(NSGetDevice(ns, &device));
  FLCHK(FlucaCGNSReadInfo(v->tmpl, N, &step, &t, &nsteps));
  for (int d = 0; d < 3; ++d)
    if (N[d] != lay.N[d]) return E_LIB; /* "Mesh size does not match CGNS zone size", cartcgns.c:697 */
  if (step < 0 || nsteps < 1) return E_LIB; /* no FlowSolution<n> / no BaseIterativeData */
  size_t big = (size_t)sz[0];
  for (int l = 0; l < 3; ++l)
    if ((size_t)sz[1 + l] > big) big = (size_t)sz[1 + l];
  double *h = (double *)malloc(sizeof(double) * (big ? big : 1)), *hf[3] = {0};
  if (!h) return E_MEM;
  FlErrorCode rc = 0;
  for (int q = 0; q < 5 && !rc; ++q) {
    rc = FlucaCGNSReadCellField(v->tmpl, &lay, step, cell_fields[q], h);
    if (!rc) rc = fl_memcpy_h2d(device, cell[q], h, sizeof(double) * (size_t)sz[0]);
  }
  for (int l = 0; l < 3 && !rc; ++l) {
    hf[l] = (double *)malloc(sizeof(double) * (size_t)(sz[1 + l] > 0 ? sz[1 + l] : 1));
    if (!hf[l]) rc = E_MEM;
  }
  if (!rc) rc = FlucaCGNSReadFaceField(v->tmpl, &lay, step, face_fields[0], hf);
  for (int l = 0; l < 3 && !rc; ++l) rc = fl_memcpy_h2d(device, face[l], hf[l], sizeof(double) * (size_t)sz[1 + l]);
  for (int l = 0; l < 3; ++l) free(hf[l]);
  free(h);
  if (!rc) rc = NSSetTimeStepAndTime(ns, step, t); /* nssol.c:199-201 */
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
