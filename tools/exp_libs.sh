# usage: tools/exp_libs.sh "<lib tags>" [bench args] : alternate prebuilt pam_amd/lib<tag>.so in one GPU call
tags=$1; shift
for rep in 1 2; do
  for v in $tags; do
    cp pam_amd/lib$v.so pam_amd/libpam_amd_awfl.so
    bash tools/exp_ab.sh "$v.$rep" "$@"
  done
done
