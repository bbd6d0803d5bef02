// snapshot_io.h -- the snapshot / IC wire format (bigfile blocks), see snapshot_io.hip
#pragma once
#include "mpg_common.h"
#include <string>

namespace mpg {
struct BigBlockInfo {
    char dtype[8];
    int nmemb, nfile;
    int64_t size;
};
void bigfile_block_info(const char *file, const char *block, BigBlockInfo *info);
void bigfile_read_block(const char *file, const char *block, int64_t start, int64_t count, const char *want_dtype, void *out);
void bigfile_write_block(const char *file, const char *block, const char *dtype, int nmemb, int nfile, int64_t size, const char *src_dtype,
                         const void *data);
int bigfile_get_attr(const char *file, const char *block, const char *name, const char *want_dtype, void *out, int nmemb);
void bigfile_set_attr(const char *file, const char *block, const char *name, const char *dtype, const void *data, int nmemb);
} // namespace mpg
