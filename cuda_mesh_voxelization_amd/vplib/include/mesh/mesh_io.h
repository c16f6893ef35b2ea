// mesh_io.h -- OBJ import/export (/root/reference/vplib/src/mesh/mesh_io.h:15,24).
#ifndef VPLIB_MESH_IO_H
#define VPLIB_MESH_IO_H

#include <string>

#include "mesh/mesh.h"

// Accepts what the reference accepts: "v x y z [r g b]", "vn x y z", "f a//b c//d e//f" (1-based).
// With VPLIB_MESH_CACHE=1 in the environment the parsed arrays are kept in "<file>.vpmesh" next to the source and reloaded
// (keyed on the source's size and modification time) instead of re-parsing -- for the 10-million-triangle inputs.
bool ImportMesh(const std::string filename, Mesh& mesh);
bool ExportMesh(const std::string filename, const Mesh& mesh);

#endif
