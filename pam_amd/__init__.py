"""pam_amd: MI355X-native AWFL dycore step behind PAM's Dycore / PamCoupler plug-in surface."""
